"""CPU tests: the oracle (oracle/al_oracle.c) against the golden vectors the REFERENCE build produced
(tests/golden/make_goldens.py).  This is what pins the oracle."""
import hashlib
import json
import os
import subprocess

import pytest

SETS_TAPS = ["g1_mt150pe", "g2_100se", "g2_250pe", "g3_adversarial", "g4_q_inv", "g4_q2"]
SETS_ALL = SETS_TAPS + ["g4_MT_orang", "g6_repeats"]


def _meta(d):
    return json.load(open(os.path.join(d, "meta.json")))


def _run(oracle_bin, d, extra):
    m = _meta(d)
    cmd = [oracle_bin] + extra
    if m.get("rg") and "--count" not in extra:
        cmd += ["-R", m["rg"]]
    return subprocess.run(cmd + [m["ref"]] + m["reads"], cwd=d, capture_output=True, check=True)


@pytest.mark.parametrize("name", SETS_ALL)
def test_sam_identical(oracle_bin, golden_unpacked, name):
    d = golden_unpacked[name]
    got = _run(oracle_bin, d, []).stdout
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert hashlib.md5(exp).hexdigest() == _meta(d)["sam_md5"]
    assert got == exp


@pytest.mark.parametrize("name", SETS_TAPS)
def test_seed_and_chain_taps(oracle_bin, golden_unpacked, name):
    d = golden_unpacked[name]
    got = _run(oracle_bin, d, ["--seeds"]).stderr
    assert got == open(os.path.join(d, "expected.seeds"), "rb").read()


@pytest.mark.parametrize("name", SETS_TAPS)
def test_ksw_taps(oracle_bin, golden_unpacked, name):
    d = golden_unpacked[name]
    got = _run(oracle_bin, d, ["--alnseq"]).stderr
    assert got == open(os.path.join(d, "expected.alnseq"), "rb").read()


@pytest.mark.parametrize("name", ["g1_mt150pe", "g2_100se", "g2_250pe", "g3_adversarial", "g6_repeats"])
def test_alser_count(oracle_bin, golden_unpacked, name):
    """a8: the as-shipped fork's only observable (map.c:299-312, main.c:417)."""
    d = golden_unpacked[name]
    got = int(_run(oracle_bin, d, ["--count"]).stdout)
    assert got == _meta(d)["count"]


def test_yeast100k_digest(oracle_bin, tmp_path):
    """G5: 100 k pairs on the 12 Mbp synthetic genome; only the digest of SAM columns 1-9 is committed."""
    import gen_synth
    from conftest import GOLDEN
    m = json.load(open(os.path.join(GOLDEN, "g5_yeast100k", "meta.json")))
    gen_synth.generate(m["config"], str(tmp_path), pairs=m["pairs"], seed=m["seed"])
    sam = subprocess.run([oracle_bin, "-t", "8", "ref.fa", "reads_1.fq", "reads_2.fq"], cwd=tmp_path, capture_output=True, check=True).stdout
    cols = b"\n".join(b"\t".join(l.split(b"\t")[:9]) for l in sam.split(b"\n") if l and not l.startswith(b"@"))
    assert cols.count(b"\n") + 1 == m["n_records"]
    assert hashlib.md5(cols).hexdigest() == m["cols1_9_md5"]
    assert hashlib.md5(sam).hexdigest() == m["sam_md5"]
