#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / mean duration.
usage: rocpd_summary.py results.db [top_n] [--last-ms X] | results.db --dump <name substring> [limit]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
    cur = sqlite3.connect(db).cursor()
    where, note = "", ""
    if "--last-ms" in sys.argv:       # only dispatches in the last X ms of the trace (steady state)
        ms = float(sys.argv[sys.argv.index("--last-ms") + 1])
        t1 = list(cur.execute("select max(end) from kernels"))[0][0]
        where, note = f" where start >= {t1 - int(ms * 1e6)}", f" (last {ms:g} ms of the trace)"
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), "
                            f"max(end-start) from kernels{where} group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}{note}")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    for name, n, tot, avg, mn, mx in rows[:top]:
        print(f"{tot / 1e6:10.3f} {100 * tot / total:6.2f} {n:6d} {avg / 1e3:10.1f} {mn / 1e3:10.1f} "
              f"{mx / 1e3:10.1f}  {name[:150]}")


def dump(db, pattern, limit=80):
    """Per-dispatch durations (launch order) of kernels whose name matches `pattern`."""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    gy = "grid_size_y" if "grid_size_y" in cols else ("grid_y" if "grid_y" in cols else None)
    sel = "name, end-start" + (f", {gx}, {gy}" if gx and gy else "")
    rows = list(cur.execute(f"select {sel} from kernels where name like ? order by start", (f"%{pattern}%",)))
    print(f"# per-dispatch: {len(rows)} launches matching {pattern!r}; columns {cols}")
    for r in rows[-limit:]:
        print(f"{r[1] / 1e3:9.1f} us  grid {r[2:] if len(r) > 2 else ''}  {r[0][:60]}")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--dump":
        dump(sys.argv[1], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 80)
        sys.exit(0)
    main()
