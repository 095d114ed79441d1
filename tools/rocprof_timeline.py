#!/usr/bin/env python
"""Print the GPU timeline (kernel start/end, gaps) of the last N ms of a rocprofv3 rocpd database."""
import sqlite3
import sys


def main(path, window_ms=12.0):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    t_end = rows[-1][2]
    rows = [r for r in rows if r[1] >= t_end - window_ms * 1e6]
    t0 = rows[0][1]
    prev_end = t0
    busy = 0
    for name, st, en in rows:
        gap = (st - prev_end) / 1e3
        busy += (en - st)
        print(f"{(st - t0) / 1e3:10.1f} us  +gap {gap:8.1f}  dur {(en - st) / 1e3:9.1f}  {name[:90]}")
        prev_end = max(prev_end, en)
    print(f"# window {(prev_end - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")


def one_step(path, marker="gp_adam_multi_kernel", which=12):
    """Kernels between the which-th and (which+1)-th launch of `marker` (one train step in steady state)."""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = marks[which], marks[which + 1]
    prev_end = rows[a][2]
    t0 = prev_end
    busy = 0
    for name, st, en in rows[a + 1:b + 1]:
        busy += en - st
        print(f"{(st - t0) / 1e3:10.1f} us  +gap {(st - prev_end) / 1e3:8.1f}  dur {(en - st) / 1e3:9.1f}  {name[:90]}")
        prev_end = max(prev_end, en)
    print(f"# step {(prev_end - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "step":
        one_step(sys.argv[1], which=int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    else:
        main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 12.0)
