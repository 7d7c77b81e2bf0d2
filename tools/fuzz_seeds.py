#!/usr/bin/env python3
"""Run given fuzz seeds through both schedules (dev tool): fuzz_seeds.py seed [seed ...]"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import fuzz_parity, oracle_lib
from goldpolish_amd import core
O = oracle_lib.Oracle(str(ROOT / "oracle" / "libgpo.so"))
with core.Context(0) as ctx:
    for seed in map(int, sys.argv[1:]):
        for mode in ("0", "1", "2"):
            print(seed, mode, fuzz_parity.run_case(core, ctx, O, seed, mode) or "ok", flush=True)
