#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace rocpd database (r*_results.db) into a per-kernel table:
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_<what>.txt"""
import sqlite3
import sys


def main(path, top=60):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"# total kernel time {tot:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  name")
    for r in rows[:top]:
        print(f"{r[2]:10.3f} {100 * r[2] / tot:6.2f} {r[1]:6d} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f}  {r[0][:120]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
