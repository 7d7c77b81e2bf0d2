#!/usr/bin/env python3
"""Regenerate tests/golden/nthash_kat.json from the REFERENCE's own ntHash header.

Authoring-container only: builds oracle/_ref/nthash_ref (oracle/Makefile `ref`
target compiles oracle/ref_nthash_driver.cpp against
/root/reference/subprojects/ntedit/lib/nthash.hpp where it lies) and stores its
output.  The JSON is data (inputs + expected hashes), committed so the GPU box
and later rounds can pin the oracle and the HIP kernels without the reference.
"""
import json
import pathlib
import subprocess

root = pathlib.Path(__file__).resolve().parents[2]
subprocess.check_call(["make", "-C", str(root / "oracle"), "ref"])
out = subprocess.check_output([str(root / "oracle" / "_ref" / "nthash_ref")])
vec = json.loads(out)
(root / "tests" / "golden" / "nthash_kat.json").write_text(json.dumps(vec, separators=(",", ":")))
print(f"wrote {len(vec)} vectors")
