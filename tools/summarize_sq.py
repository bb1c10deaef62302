#!/usr/bin/env python3
"""Condense the rocpd databases of tools/profile_sq.sh into a text table. Usage: summarize_sq.py gpurun_out/<tag> > out"""
import sqlite3, sys, os
root = sys.argv[1]
print("# SQ counters of the RTI kernel (nmpc::rti_block_kernel since round 3), per launch (mean of 25 eager launches), rocprofv3 --pmc in two passes (tools/profile_sq.sh).")
print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles).\n")
for B in (4096, 32768):
    per_kernel = {}
    for p in ("p1", "p2"):
        db = os.path.join(root, f"sq_B{B}", p, f"{p}_results.db")
        if not os.path.exists(db):
            continue
        c = sqlite3.connect(db)
        for kname, name, avg, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                             "where kernel_name like '%rti_%kernel%' group by kernel_name, counter_name"):
            d = per_kernel.setdefault(kname.split("(")[0].replace("void ", ""), {})
            d[name] = avg
            d["_launches"] = n
    # one process runs the step loop twice: launches in flight (lane mapping packed for all of them) and in order
    for kname, vals in sorted(per_kernel.items()):
        w = vals["SQ_WAVES"]
        ppw = B / w
        print(f"B = {B} (N = 20), {kname}: {int(w)} wavefronts, {ppw:.0f} problems per wavefront, {vals['_launches']} launches")
        for k in sorted(vals):
            if not k.startswith("_"):
                print(f"  {k:22s} {vals[k]:16.1f}   per wavefront {vals[k] / w:10.1f}")
        print(f"  -> VALU instructions per wavefront {vals['SQ_INSTS_VALU'] / w:.0f} ({vals['SQ_INSTS_VALU'] / w / ppw:.0f} per problem), "
              f"wavefront lifetime {4 * vals['SQ_WAVE_CYCLES'] / w:.0f} cycles: issuing {100 * vals['SQ_ACTIVE_INST_ANY'] / vals['SQ_WAVE_CYCLES']:.0f} %, "
              f"parked at s_waitcnt/barrier {100 * vals['SQ_WAIT_ANY'] / vals['SQ_WAVE_CYCLES']:.0f} %, issue-stalled {100 * vals['SQ_WAIT_INST_ANY'] / vals['SQ_WAVE_CYCLES']:.0f} %\n")
