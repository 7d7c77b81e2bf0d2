#!/usr/bin/env python3
"""debug aid: gp_sealer_consensus on slices of tests/golden/nw_kat.json (mode = pairs:a:b | multi:a:b | str:A,B)"""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import core
kat = json.loads((ROOT / "tests/golden/nw_kat.json").read_text())
mode = sys.argv[1].split(":")
if mode[0] == "pairs":
    vs = kat["pairs"][int(mode[1]):int(mode[2])]
    sets, want = [[v["a"], v["b"]] for v in vs], [v["consensus"] for v in vs]
elif mode[0] == "multi":
    vs = kat["multi"][int(mode[1]):int(mode[2])]
    sets, want = [v["seqs"] for v in vs], [v["consensus"] for v in vs]
else:
    sets, want = [mode[1].split(",")], None
t = time.time()
with core.Context(0) as ctx:
    got = core.sealer_consensus(ctx, sets)
print(sys.argv[1], "ok" if want is None or [g.decode() for g in got] == want else "MISMATCH", round(time.time() - t, 3), "s",
      [g.decode()[:40] for g in got[:2]], flush=True)
