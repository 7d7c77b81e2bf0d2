#!/usr/bin/env python3
"""Dev tool: run one test against another build of the library (e.g. the previous round's, kept as
goldpolish_amd/libgpcore_old.so) and report whether it fails there -- shows that a new test
really catches the bug it was written for.  old_lib_check.py <lib> <pytest node id>; exit 0 = the
test FAILED with that library."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import pytest
from goldpolish_amd import core

core.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
rc = pytest.main(["-x", "-q", "-m", "gpu", sys.argv[2]])
print("pytest exit code with", core.LIB_PATH.name, "=", int(rc))
sys.exit(0 if int(rc) == 1 else 1)
