"""Turn a rocprofv3 results .db (ROCm 7.2 default output) into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` prints, plus LDS / scratch use.  (The trace's `vgpr_count` / `accum_vgpr_count` columns are NOT the
allocation - they read 136 for a kernel the compiler allocates 256 + 12 registers for - so they are not printed: the allocation is in
profiles/r*_kernel_resources.json, from the compiler's own resource report, tools/kernel_resources.py.)  Usage:
    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db > profiles/r01_x_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-72s %6s %12s %12s %12s %12s %6s %5s %7s %7s %8s %5s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "sgpr", "lds_B", "scr_B", "grid", "wg"))
    for r in rows:
        print("%-72s %6d %12.1f %12.2f %12.2f %12.2f %6.2f %5d %7d %7d %8d %5d" % (
            r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))


if __name__ == "__main__":
    main(sys.argv[1])
