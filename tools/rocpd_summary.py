#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) as a per-kernel table:

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x.txt

Also prints, when counters were collected (--pmc), the per-kernel mean of each.
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 kernel-trace summary of %s" % path)
    print("%-64s %7s %12s %10s %10s %10s %6s %9s %5s %5s %5s %7s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid_x", "wg_x", "vgpr", "sgpr", "lds"))
    for name, calls, tot, avg, mn, mx, gx, wx, vg, sg, lds in rows:
        short = name if len(name) <= 64 else name[:61] + "..."
        print("%-64s %7d %12.1f %10.2f %10.2f %10.2f %6.1f %9d %5d %5d %5d %7d" % (
            short, calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, gx, wx, vg or 0, sg or 0, lds or 0))
    try:
        pmc = cur.execute(
            "select name, counter_name, avg(counter_value), count(*) from pmc_events "
            "group by name, counter_name order by name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# counters (mean per dispatch; FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 derives them;"
              "\n# on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md section HBM)")
        for name, counter, val, cnt in pmc:
            short = name if len(name) <= 64 else name[:61] + "..."
            print("%-64s %-28s %16.1f  (n=%d)" % (short, counter, val, cnt))


if __name__ == "__main__":
    main(sys.argv[1])
