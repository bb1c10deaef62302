cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wbtr -o t -- python3 tools/wb_rows_time.py > gpurun_out/wbtr.log 2>&1
python3 - <<PY
import glob,csv
fs=glob.glob("gpurun_out/wbtr/**/*kernel_stats.csv", recursive=True)
print(fs)
for f in fs:
    for r in csv.DictReader(open(f)):
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
tail -3 gpurun_out/wbtr.log
rm -rf gpurun_out/wbtr
