#!/usr/bin/env python3
"""Condense the outputs of tools/profile_r04.sh (gpurun_out/<tag>/) into the text / JSON records kept under profiles/.
Runs on the GPU box right after the passes (the rocpd databases stay there):  summarize_r04.py <out_dir> <tag>"""
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


for name, d in (("kernel_stats", "trace"), ("kernel_stats_driver_flags", "trace_driver"), ("wb_kernel_stats", "wb_trace"), ("ltv_kernel_stats", "ltv_trace"),
                ("backend_kernel_stats", "be_trace"), ("extras_kernel_stats", "extras_trace")):
    if os.path.isdir(os.path.join(out, d)):
        subprocess.run([sys.executable, os.path.join(here, "summarize_prof.py"), os.path.join(out, d), os.path.join(dst, f"{tag}_{name}.txt")])
# ---- counters of the RTI kernels: every launch of the many-batch build served BPL batches of B problems
BPL, B, N = 200, 4096, 20
import sqlite3 as _sq


def counter_means(dirs):
    per = {}
    for d in dirs:
        for db in glob.glob(os.path.join(out, d, "**", "*_results.db"), recursive=True):
            c = _sq.connect(db)
            for kname, cname, avg, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                                  "where kernel_name like '%rti_%kernel%' group by kernel_name, counter_name"):
                k = per.setdefault(kname.split("(")[0].replace("void ", ""), {})
                k[cname] = avg
                k["_launches"] = n
    return per


lines = ["# SQ counters of nmpc::rti_block_kernel, mean per launch, rocprofv3 --pmc in two passes over",
         f"#   bench.py --no-graph --steps {BPL} --warmup {BPL}: one grid of the <4, 5, ..., FULLN> build = {BPL} batches of {B} problems (alore_nmpc_rti_many),",
         "#   one grid of the <16, 2> build = one batch (the in_order pass).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles).", ""]
sq = counter_means(("sq_p1", "sq_p2"))
valu_pp = {}
for kname, v in sorted(sq.items()):
    if "SQ_WAVES" not in v:
        continue
    w = v["SQ_WAVES"]
    many = "<4, 5" in kname
    problems = B * (BPL if many else 1)
    lines.append(f"{kname}: {int(w)} wavefronts per launch, {problems / w:.0f} problems per wavefront, {v['_launches']} launches")
    for k in sorted(v):
        if not k.startswith("_"):
            lines.append(f"  {k:22s} {v[k]:18.1f}   per wavefront {v[k] / w:10.1f}")
    if "SQ_INSTS_VALU" in v and "SQ_WAVE_CYCLES" in v:
        valu_pp[kname] = v["SQ_INSTS_VALU"] / problems
        act = 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
        lines.append(f"  -> VALU instructions per wavefront {v['SQ_INSTS_VALU'] / w:.0f} = {v['SQ_INSTS_VALU'] / problems:.0f} per problem; wavefront lifetime "
                     f"{4 * v['SQ_WAVE_CYCLES'] / w:.0f} cycles: issuing {act:.0f} %, parked at s_waitcnt/barrier {100 * v.get('SQ_WAIT_ANY', 0) / v['SQ_WAVE_CYCLES']:.0f} %, "
                     f"issue-stalled {100 * v.get('SQ_WAIT_INST_ANY', 0) / v['SQ_WAVE_CYCLES']:.0f} %")
    lines.append("")
open(os.path.join(dst, f"{tag}_sq_counters.txt"), "w").write("\n".join(lines) + "\n")
tr = counter_means(("pmc_fetch", "pmc_write"))
rec = {}
for kname, v in tr.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    many = "<4, 5" in kname
    per_batch = BPL if many else 1
    f_kb, w_kb = v["FETCH_SIZE"] / per_batch, v["WRITE_SIZE"] / per_batch
    rec[kname] = {"bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024), "lower_bound_bytes": int((f_kb + w_kb) * 1024),
                  "raw": {"FETCH_SIZE_KB_per_batch": round(f_kb, 2), "WRITE_SIZE_KB_per_batch": round(w_kb, 2), "launches": v["_launches"],
                          "batches_per_launch": per_batch}}
main_k = [k for k in rec if "<4, 5" in k]
if main_k:
    e = rec[main_k[0]]
    json.dump({f"B{B}_N{N}": {
        "bytes_per_launch": e["bytes_per_launch"], "raw": e["raw"], "algorithmic_bytes_per_launch": 4192 * B,
        "kernel": "nmpc::rti_block_kernel", "instantiation": main_k[0],
        "one_batch_at_a_time": {k: v for k, v in rec.items() if k not in main_k},
        "source": f"profiles/{tag}_* (tools/profile_r05.sh / profile_r04.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, eager launches)",
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --no-graph --steps 200 --warmup 200: every launch of "
               "the many-batch build serves 200 batches of B problems, the per-launch mean is divided by 200 (`bytes_per_launch` = bytes of ONE B-problem "
               "batch, the unit of roofline.achieved). bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 reports half the bytes of 16-byte-per-lane "
               "streaming reads (MI355X_MICROARCH.md, HBM section). 76 % of the input bytes (W, y by LDS-DMA; bounds) are read 16 B/lane, the rest "
               "(x, u, od, dual) in 8- and 12-byte pieces, for which the x2 over-corrects: the true figure lies between lower_bound "
               "(FETCH+WRITE)*1024 and this number; the single-iteration build reads x and u a second time at the end (L2 / MALL)."}},
        open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
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
for f in ("bench.json", "bench_driver_flags.json", "bench_whole_body.json", "bench_under_rocprof.json", "bench_driver_flags_under_rocprof.json", "wb_run.txt", "ltv_run.txt", "be_run.txt"):
    p = os.path.join(out, f)
    if os.path.exists(p):
        open(os.path.join(dst, f"{tag}_{f}"), "w").write(open(p).read())
print("\n".join(sorted(os.listdir(dst))))
