#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short text
summary for profiles/ (our kernels in full, everything else as one line)."""
import csv, glob, os, sys

def main(d, out):
    stats = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)
    trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    lines = []
    for f in stats:
        rows = list(csv.DictReader(open(f)))
        lines.append(f"# {os.path.basename(f)}")
        lines.append("name,calls,total_ns,avg_ns,pct,min_ns,max_ns,stddev")
        other = 0
        for r in rows:
            if "nmpc::" in r["Name"]:
                lines.append(",".join([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                       r["MinNs"], r["MaxNs"], r["StdDev"]]))
            else:
                other += int(r["TotalDurationNs"])
        lines.append(f"(all non-nmpc kernels: torch fills/copies of the bench setup),,{other},,,,,")
    for f in trace:
        rows = [r for r in csv.DictReader(open(f)) if "rti_kernel" in r["Kernel_Name"] or "nmpc::" in r["Kernel_Name"]]
        seen = set()
        lines.append(f"# {os.path.basename(f)}: launch geometry / resources per nmpc kernel")
        for r in rows:
            key = (r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"])
            if key in seen:
                continue
            seen.add(key)
            lines.append(f'{r["Kernel_Name"]}: grid={r["Grid_Size_X"]} wg={r["Workgroup_Size_X"]} '
                         f'VGPR={r["VGPR_Count"]} AGPR={r["Accum_VGPR_Count"]} SGPR={r["SGPR_Count"]} '
                         f'LDS_static={r["LDS_Block_Size"]} scratch={r["Scratch_Size"]}')
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
