#!/usr/bin/env python3
"""Per-kernel means of every counter in rocprofv3 --pmc result databases (summed over the dimension instances rocprofv3
reports per dispatch).  Usage: summarize_counters.py <dir> [<dir> ...] <kernel substring> <out.txt>"""
import glob, os, sqlite3, sys


def main(dirs, kern, out):
    lines = [f"# counters of kernels matching '{kern}' (per-dispatch totals over all instances, mean over dispatches)"]
    agg = {}
    for d in dirs:
        for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
            c = sqlite3.connect(db)
            rows = c.execute("select kernel_name, dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
                             "where kernel_name like ? group by kernel_name, dispatch_id, counter_name", (f"%{kern}%",)).fetchall()
            for name, disp, cn, val, dur in rows:
                a = agg.setdefault((name, cn), [0.0, 0, 0.0])
                a[0] += val; a[1] += 1; a[2] += dur
    for (name, cn), (tot, n, dur) in sorted(agg.items()):
        lines.append(f"{name.split('(')[0]:40s} {cn:32s} mean {tot / n:16.1f}  dispatches {n:3d}  mean duration {dur / n / 1e3:9.1f} us")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1:-2], sys.argv[-2], sys.argv[-1])
