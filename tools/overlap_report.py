#!/usr/bin/env python3
"""How much of the filter build (stage A) ran beside the kernels behind it (stages B and C).

Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints, for the traced process: the
wall time between the first kernel start and the last kernel end, the time during which at least
one stage-A kernel was running, the same for stages B/C, and the time both were running.

    python tools/overlap_report.py gpurun_out/prof/*/*_kernel_trace.csv
"""
import csv
import sys

A_PREFIXES = ("cbf_", "census_", "hash_", "stage_a")


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def main(path):
    a, bc, per = [], [], {}
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].split("::")[-1]
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            (a if name.startswith(A_PREFIXES) else bc).append((s, e))
            t = per.setdefault(name, [0, 0])
            t[0] += 1
            t[1] += e - s
    ua, ub = union(a), union(bc)
    both = intersect(ua, ub)
    t0 = min(x[0] for x in a + bc)
    t1 = max(x[1] for x in a + bc)
    ms = 1e-6
    print(f"wall {ms * (t1 - t0):.1f} ms; stage A busy {ms * length(ua):.1f}; B/C busy {ms * length(ub):.1f}; "
          f"both {ms * length(both):.1f}; neither {ms * ((t1 - t0) - length(ua) - length(ub) + length(both)):.1f}")
    for name, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {name:28s} {n:7d} launches {ms * t:10.1f} ms")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(p)
        main(p)
