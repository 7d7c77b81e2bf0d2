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

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
# MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly half the bytes of a WIDE COALESCED
# STREAMING read; random 1-byte loads are reported at their 64-byte line (calibrated with
# tools/probes/random_rmw_probe: profiles/r02_probe_pmc_traffic.json).  These kernels stream their
# input (records, bitmaps, sequence tiles): their FETCH_SIZE is doubled in "FETCH_SIZE_corrected".
STREAMING = ("gp::census_mark", "gp::census_pack", "gp::ckc_scatter", "gp::ckc_group", "gp::ckc_share", "gp::ckc_apply", "gp::ntedit_scan", "gp::materialize_kernel", "gp::mask_kernel",
             "gp::upper_kernel", "gp::copy_contigs", "gp::copy_flagged", "gp::pack_kernel", "gp::classify_kernel")
raw = pathlib.Path("gpurun_out/profiles_raw")
out = pathlib.Path("profiles")
out.mkdir(exist_ok=True)
def newest(pattern):
    """gpurun merges every call's files into the same directories: only the latest run counts"""
    files = sorted(glob.glob(pattern), key=lambda f: pathlib.Path(f).stat().st_mtime)
    return files[-1:]


for name in ("config3", "config2", "survey3", "probe"):
    for f in newest(str(raw / f"{name}_trace" / "*" / "*kernel_stats.csv")):
        shutil.copy(f, out / f"{rnd}_{name}_kernel_stats.csv")
b3 = raw / "survey3_bench.json"
if b3.exists():
    shutil.copy(b3, out / f"{rnd}_survey3_bench_line.json")
for name in ("config3", "config2", "probe"):
    traffic = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in newest(str(raw / f"{name}_{kind}" / "*" / "*counter_collection.csv")):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                k = r["Kernel_Name"].split("(")[0]
                if k.startswith("void gp::census_"):      # templates over the record geometry: one name per kernel
                    k = k[len("void "):].split("<")[0]
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
            for k, (n, v) in agg.items():
                traffic.setdefault(k, {})[counter] = {"launches": n, "sum_KB": round(v, 1),
                                                      "per_launch_KB": round(v / n, 1)}
    for k, t in traffic.items():
        if k in STREAMING and "FETCH_SIZE" in t:
            t["FETCH_SIZE_corrected"] = {"per_launch_KB": round(2 * t["FETCH_SIZE"]["per_launch_KB"], 1),
                                         "sum_KB": round(2 * t["FETCH_SIZE"]["sum_KB"], 1),
                                         "why": "wide coalesced streaming read: x2 (MI355X_MICROARCH.md, HBM)"}
    (out / f"{rnd}_{name}_pmc_traffic.json").write_text(json.dumps(traffic, indent=1, sort_keys=True))
    b = raw / f"{name}_bench.json"
    if b.exists():
        shutil.copy(b, out / f"{rnd}_{name}_bench_line.json")
    print(name, "kernels with PMC data:", len(traffic))
