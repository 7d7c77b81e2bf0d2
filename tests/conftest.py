import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libgpo.so), built on demand with gcc."""
    so = ROOT / "oracle" / "libgpo.so"
    if not so.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "libgpo.so"])
    import oracle_lib

    return oracle_lib.Oracle(str(so))
