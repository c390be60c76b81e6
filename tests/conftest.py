import gzip
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The goldens' @PG line has no VN:/CL: (the reference driver that made them has no argv to pass, oracle/ref_driver.c): the CLI
# is run with the bare line except in the test that pins VN:/CL: against the fork's own main() (test_pg_line_*).
os.environ.setdefault("AL_PG_PLAIN", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """Builds the CPU oracle CLI (test infrastructure) and returns its path."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "al_oracle", "libal_oracle.so"], check=True)
    return os.path.join(ROOT, "oracle", "al_oracle")


@pytest.fixture(scope="session")
def golden_unpacked(tmp_path_factory):
    """Unpacks tests/golden/*/ *.gz into a temp dir; returns {set: dir}."""
    out = {}
    base = tmp_path_factory.mktemp("golden")
    for name in sorted(os.listdir(GOLDEN)):
        d = os.path.join(GOLDEN, name)
        if not os.path.isdir(d):
            continue
        o = base / name
        o.mkdir()
        for fn in os.listdir(d):
            src = os.path.join(d, fn)
            if fn.endswith(".gz"):
                (o / fn[:-3]).write_bytes(gzip.open(src, "rb").read())
            else:
                (o / fn).write_bytes(open(src, "rb").read())
        out[name] = str(o)
    return out
