#!/usr/bin/env python3
"""Per-launch HBM bytes of nmpc::rti_kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts half the
bytes of 16-byte-per-lane streaming reads).  Usage: summarize_pmc.py <fetch_dir> <write_dir> <key> <algorithmic_bytes>
Prints the JSON entry for profiles/hbm_traffic.json."""
import csv, glob, json, os, sys


KERNEL_IS_BLOCK = [False]
SOURCE = [None]


def mean_counter(d, name):
    vals = []
    for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):  # rocpd output
        import sqlite3
        c = sqlite3.connect(db)
        vals += [float(v) for (v,) in c.execute(
            "select value from counters_collection where kernel_name like '%rti_%kernel%' and counter_name = ?", (name,))]
        if c.execute("select count(*) from counters_collection where kernel_name like '%rti_block_kernel%'").fetchone()[0]:
            KERNEL_IS_BLOCK[0] = True
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ("rti_kernel" in r["Kernel_Name"] or "rti_block_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {name} rows for rti_kernel under {d}")
    return sum(vals) / len(vals), len(vals)


def main(fetch_dir, write_dir, key, alg, source=None):
    SOURCE[0] = source
    f_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE")
    w_kb, nw = mean_counter(write_dir, "WRITE_SIZE")
    entry = {
        "bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024),
        "raw": {"FETCH_SIZE_KB": round(f_kb, 2), "WRITE_SIZE_KB": round(w_kb, 2), "launches": min(nf, nw)},
        "algorithmic_bytes_per_launch": int(alg),
        "kernel": "nmpc::rti_block_kernel" if KERNEL_IS_BLOCK[0] else "nmpc::rti_kernel",
        "source": SOURCE[0],
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py (eager launches); "
               "per-launch mean of the RTI kernel this record names. bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 reports "
               "half the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section). 76 % of the input bytes "
               "(W, y, bounds) are read 16 B/lane, the rest (x, u, od, dual) in 8- and 12-byte pieces, for which the x2 over-corrects: "
               "the true figure lies between (FETCH+WRITE)*1024 and this number; the kernel reads every input byte "
               "exactly once (W, y by LDS-DMA, 16 B per lane; the iterate, od, bounds and dual in 8- and 12-byte "
               "pieces per lane since round 3).",
    }
    print(json.dumps({key: entry}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
