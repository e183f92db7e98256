#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / mean duration.
usage: rocpd_summary.py results.db [top_n]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), "
                            "max(end-start) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  kernel")
    for name, n, tot, avg, mn, mx in rows[:top]:
        print(f"{tot / 1e6:10.3f} {100 * tot / total:6.2f} {n:6d} {avg / 1e3:10.1f} {mn / 1e3:10.1f} "
              f"{mx / 1e3:10.1f}  {name[:150]}")


if __name__ == "__main__":
    main()
