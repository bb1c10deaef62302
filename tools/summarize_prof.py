#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a short text
summary for profiles/ (our kernels in full, everything else as one line)."""
import csv, glob, os, sys

OURS = ("nmpc::", "wb::", "ltv::", "backend::")


def ours(name):
    return any(k in name for k in OURS)


def from_rocpd(db, lines):
    """rocprofv3 >= 7 writes a rocpd SQLite database unless --output-format csv is given."""
    import sqlite3
    c = sqlite3.connect(db)
    lines.append(f"# {os.path.basename(db)} (rocpd): per-kernel statistics, durations in ns")
    lines.append("name,calls,total_ns,avg_ns,min_ns,max_ns")
    other = 0
    for name, n, tot, avg, mn, mx in c.execute(
            "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name"):
        if ours(name):
            lines.append(f"{name},{n},{tot},{avg:.1f},{mn},{mx}")
        else:
            other += tot
    lines.append(f"(all non-nmpc kernels: torch fills/copies of the bench setup),,{other},,,")
    # the RTI kernel serves a whole group of batches per launch (alore_nmpc_rti_many): per launch geometry, so that the grid of
    # the timed region (K batches) can be told from warm-up / steady-state / spot-check launches of the same kernel
    lines.append("# nmpc::rti_* per launch geometry: name,grid_x,calls,avg_ns,min_ns,max_ns")
    for name, gx, n, avg, mn, mx in c.execute(
            "select name, grid_x, count(*), avg(end-start), min(end-start), max(end-start) from kernels where name like '%nmpc::rti_%' "
            "group by name, grid_x order by name, grid_x"):
        lines.append(f"{name},{gx},{n},{avg:.1f},{mn},{mx}")
    # every launch of the grid build in launch order (a bench.py run: 5-batch warm-up grids, the 200-batch steady-state grid, then the
    # TIMED K-batch grid, the same grid again for the parity spot check and once more on the stress distribution)
    lines.append("# nmpc::rti_block_kernel<4, 5, ...> launches in order: grid_x,duration_ns")
    for gx, dur in c.execute("select grid_x, end-start from kernels where name like '%nmpc::rti_block_kernel<4, 5%' order by start"):
        lines.append(f"{gx},{dur}")
    lines.append("# launch geometry / resources per nmpc kernel")
    for r in c.execute("select distinct name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size "
                       "from kernels"):
        if not ours(r[0]):
            continue
        lines.append(f"{r[0]}: grid={r[1]} wg={r[2]} VGPR={r[3]} AGPR={r[4]} SGPR={r[5]} LDS={r[6]} scratch={r[7]}")


def main(d, out):
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if dbs:
        lines = []
        for db in dbs:
            from_rocpd(db, lines)
        open(out, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
        return
    stats = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)
    trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    lines = []
    for f in stats:
        rows = list(csv.DictReader(open(f)))
        lines.append(f"# {os.path.basename(f)}")
        lines.append("name,calls,total_ns,avg_ns,pct,min_ns,max_ns,stddev")
        other = 0
        for r in rows:
            if ours(r["Name"]):
                lines.append(",".join([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                       r["MinNs"], r["MaxNs"], r["StdDev"]]))
            else:
                other += int(r["TotalDurationNs"])
        lines.append(f"(all non-nmpc kernels: torch fills/copies of the bench setup),,{other},,,,,")
    for f in trace:
        rows = [r for r in csv.DictReader(open(f)) if ours(r["Kernel_Name"])]
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
