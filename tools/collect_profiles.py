#!/usr/bin/env python3
"""Condense gpurun_out/profiles_raw (tools/make_profiles.sh) into profiles/rNN_*: the
rocprofv3 kernel stats tables as they are, and per-kernel HBM traffic from the PMC passes."""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
raw = pathlib.Path("gpurun_out/profiles_raw")
out = pathlib.Path("profiles")
out.mkdir(exist_ok=True)
def newest(pattern):
    """gpurun merges every call's files into the same directories: only the latest run counts"""
    files = sorted(glob.glob(pattern), key=lambda f: pathlib.Path(f).stat().st_mtime)
    return files[-1:]


for name in ("config3", "config2", "probe"):
    for f in newest(str(raw / f"{name}_trace" / "*" / "*kernel_stats.csv")):
        shutil.copy(f, out / f"{rnd}_{name}_kernel_stats.csv")
for name in ("config3", "config2", "probe"):
    traffic = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in newest(str(raw / f"{name}_{kind}" / "*" / "*counter_collection.csv")):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                k = r["Kernel_Name"].split("(")[0]
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
            for k, (n, v) in agg.items():
                traffic.setdefault(k, {})[counter] = {"launches": n, "sum_KB": round(v, 1),
                                                      "per_launch_KB": round(v / n, 1)}
    (out / f"{rnd}_{name}_pmc_traffic.json").write_text(json.dumps(traffic, indent=1, sort_keys=True))
    b = raw / f"{name}_bench.json"
    if b.exists():
        shutil.copy(b, out / f"{rnd}_{name}_bench_line.json")
    print(name, "kernels with PMC data:", len(traffic))
