#!/usr/bin/env python3
"""development aid: per-kernel totals of a rocprofv3 --kernel-trace run (the sqlite database rocprofv3 7.x
writes: view `kernels` with name, start, end), and with a second argument the first N dispatches in order
(a third: only those of at least that many ms).
usage: trace_kernels.py DIR_OR_DB [N [MIN_MS]]"""
import glob
import sqlite3
import sys

path = sys.argv[1]
if not path.endswith(".db"):
    path = sorted(glob.glob(path + "/**/*.db", recursive=True))[-1]
db = sqlite3.connect(path)
rows = db.execute("select name, start, end from kernels order by start").fetchall()
tot = {}
for name, s, e in rows:
    n = name.split("(")[0].replace("gp::", "")
    t = tot.setdefault(n, [0, 0.0])
    t[0] += 1
    t[1] += (e - s) / 1e6
span = (rows[-1][2] - rows[0][1]) / 1e6 if rows else 0.0
print(f"{len(rows)} dispatches over {span:.1f} ms")
for n, (cnt, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:60]:60s} n={cnt:6d} total={ms:10.2f} ms  avg={ms / cnt:9.3f} ms")
if len(sys.argv) > 2:
    t0 = rows[0][1]
    min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    rows = [r for r in rows if (r[2] - r[1]) / 1e6 >= min_ms]
    for name, s, e in rows[: int(sys.argv[2])]:
        print(f"@{(s - t0) / 1e6:10.3f} ms  {(e - s) / 1e6:9.3f} ms  {name.split('(')[0][-40:]}")
