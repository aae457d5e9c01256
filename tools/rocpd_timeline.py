#!/usr/bin/env python3
"""Timeline of the last dispatches in a rocprofv3 rocpd database: begin / end of every kernel (us, relative), the queue it ran on
and the gap to the previous dispatch of the same queue -- where an iteration's time goes BETWEEN its kernels.

    python tools/rocpd_timeline.py x_results.db [how many dispatches from the end, default 24] [skip from the end, default 40]
"""
import sqlite3
import sys


def main(path, count=24, skip=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    queue = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute("select name, start, end, %s from kernels order by start" % queue).fetchall()
    rows = rows[max(0, len(rows) - skip - count):len(rows) - skip]
    t0 = rows[0][1]
    last_end = {}
    print("# %s (columns of `kernels`: %s)" % (path, ", ".join(cols)))
    print("%-44s %6s %10s %10s %9s %9s" % ("kernel", "queue", "begin_us", "end_us", "dur_us", "gap_us"))
    for name, start, end, q in rows:
        gap = (start - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = end
        print("%-44s %6s %10.2f %10.2f %9.2f %9.2f" % (name[:44], q, (start - t0) / 1e3, (end - t0) / 1e3, (end - start) / 1e3, gap))


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:4]))
