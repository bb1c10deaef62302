#!/usr/bin/env python3
"""Condense the outputs of tools/profile_r03.sh (gpurun_out/<tag>/) into the text / JSON records kept under profiles/.
Runs on the GPU box right after the passes (the rocpd databases stay there):  summarize_r03.py <out_dir> <tag>"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

out, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
dst = os.path.join(out, "summary")
os.makedirs(dst, exist_ok=True)


def run(args, to):
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True)
    open(os.path.join(dst, to), "w").write(r.stdout + (("\n# stderr:\n" + r.stderr[-2000:]) if r.returncode else ""))


for name, d in (("kernel_stats", "trace"), ("kernel_stats_in_order", "trace_in_order"), ("wb_kernel_stats", "wb_trace"), ("ltv_kernel_stats", "ltv_trace"),
                ("backend_kernel_stats", "be_trace"), ("extras_kernel_stats", "extras_trace")):
    if os.path.isdir(os.path.join(out, d)):
        subprocess.run([sys.executable, os.path.join(here, "summarize_prof.py"), os.path.join(out, d), os.path.join(dst, f"{tag}_{name}.txt")])
run([os.path.join(here, "summarize_sq.py"), out], f"{tag}_sq_counters.txt")
src = f"profiles/{tag}_* (tools/profile_r03.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, eager launches)"
run([os.path.join(here, "summarize_pmc.py"), os.path.join(out, "pmc_fetch"), os.path.join(out, "pmc_write"), "B4096_N20",
     str(4192 * 4096), src], "hbm_traffic.json")
# whole-body counters: per kernel, per launch
lines = ["# whole-body kernels, rocprofv3 --pmc (two passes), mean per launch over the launches of tools/wb_profile.py"]
stage_valu = None
for d in ("wb_pmc", "wb_pmc2"):
    for db in glob.glob(os.path.join(out, d, "**", "*_results.db"), recursive=True):
        c = sqlite3.connect(db)
        for kname, cname, avg, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                              "where kernel_name like '%wb::%' group by kernel_name, counter_name"):
            short = kname.split("(")[0].split("::")[-1]
            lines.append(f"{short:18s} {cname:32s} {avg:18.1f}   ({n} launches)")
            if "stage_kernel" in kname and cname == "SQ_INSTS_VALU":
                stage_valu = avg
            if "stage_kernel" in kname and cname == "SQ_WAVES":
                stage_waves = avg
open(os.path.join(dst, f"{tag}_wb_counters.txt"), "w").write("\n".join(lines) + "\n")
if stage_valu:
    try:
        per = stage_valu / stage_waves
    except NameError:
        per = stage_valu / (4096 * 20)
    json.dump({"valu_instructions_per_wavefront": per,
               "source": f"profiles/{tag}_wb_counters.txt (SQ_INSTS_VALU / SQ_WAVES of wb::stage_kernel, B = 4096, N = 20)"},
              open(os.path.join(dst, "wb_stage_valu.json"), "w"), indent=1)
# back_end and LTV counters
for d, pat, name in (("be_pmc", "%backend_kernel%", "backend"), ("ltv_pmc", "%get_cmd%", "ltv")):
    rows, vals = [f"# {name}: rocprofv3 --pmc, mean per launch"], {}
    for db in glob.glob(os.path.join(out, d, "**", "*_results.db"), recursive=True):
        c = sqlite3.connect(db)
        for kname, cname, avg, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                              "where kernel_name like ? group by kernel_name, counter_name", (pat,)):
            rows.append(f"{kname.split('(')[0][-40:]:42s} {cname:24s} {avg:18.1f}   ({n} launches)")
            vals[cname] = avg
    open(os.path.join(dst, f"{tag}_{name}_counters.txt"), "w").write("\n".join(rows) + "\n")
    if name == "backend" and "SQ_INSTS_VALU" in vals and "SQ_WAVES" in vals:
        json.dump({"valu_instructions_per_wavefront": vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"], "wavefronts": vals["SQ_WAVES"],
                   "source": f"profiles/{tag}_backend_counters.txt (SQ_INSTS_VALU / SQ_WAVES of backend::backend_kernel, tools/be_profile.py)"},
                  open(os.path.join(dst, "backend_valu.json"), "w"), indent=1)
for f in ("bench.json", "bench_driver_flags.json", "bench_whole_body.json", "bench_under_rocprof.json", "bench_in_order_under_rocprof.json", "wb_run.txt", "ltv_run.txt", "be_run.txt"):
    p = os.path.join(out, f)
    if os.path.exists(p):
        open(os.path.join(dst, f"{tag}_{f}"), "w").write(open(p).read())
print("\n".join(sorted(os.listdir(dst))))
