cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in v1 v2; do
    export ALORE_NMPC_LIB=$PWD/ab/libalore_nmpc_$v.so
    echo "$v: $(python bench.py --no-cpu-baseline --no-extras --no-converged --steps 300 --warmup 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,3), "us; in_order", round(d["in_order"]["ms_per_step"]*1e3,2))')"
  done
done | tee gpurun_out/g13_ab.txt
