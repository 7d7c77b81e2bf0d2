#!/usr/bin/env python3
"""Regenerate tests/golden/nw_kat.json from the REFERENCE's own Gotoh aligner.

Authoring-container only: builds oracle/_ref/align_ref (oracle/Makefile `ref` target
links oracle/ref_align_driver.cpp with /root/reference/subprojects/sealer/
Align/alignGlobal.cc, Common/Sequence.cpp and Common/Options.cpp where they lie --
standard library only, nothing copied, no stand-ins) and stores its output.  The JSON
is data (inputs + expected consensus strings), committed so the GPU box and later
rounds can pin the oracle's and the HIP kernel's consensus step
without the reference.
"""
import json
import pathlib
import subprocess

root = pathlib.Path(__file__).resolve().parents[2]
subprocess.check_call(["make", "-C", str(root / "oracle"), "ref"])
out = subprocess.check_output([str(root / "oracle" / "_ref" / "align_ref")])
vec = json.loads(out)
(root / "tests" / "golden" / "nw_kat.json").write_text(json.dumps(vec, separators=(",", ":")))
print({k: len(v) for k, v in vec.items()})
