"""-m gpu parity tests, end to end: the `airlift-align` drop-in (HIP path behind the C-ABI) must print byte-identical
SAM to what the reference build printed for the golden inputs (CIGAR/POS/MAPQ/flags/tags bit-exact)."""
import hashlib
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
SETS = ["g1_mt150pe", "g2_100se", "g2_250pe", "g3_adversarial", "g4_q2", "g6_repeats"]


def _diff_report(got, exp, name):
    g, e = got.split(b"\n"), exp.split(b"\n")
    out = ["%s: %d vs %d lines" % (name, len(g), len(e))]
    nbad = 0
    for i in range(min(len(g), len(e))):
        if g[i] != e[i]:
            nbad += 1
            if nbad <= 8:
                gf, ef = g[i].split(b"\t"), e[i].split(b"\t")
                cols = [j for j in range(min(len(gf), len(ef))) if gf[j] != ef[j]]
                out.append("line %d cols %s\n  got %s\n  exp %s" % (i, cols, b"\t".join(gf[:9] + gf[11:]).decode(), b"\t".join(ef[:9] + ef[11:]).decode()))
    out.append("%d differing lines" % nbad)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "samdiff_%s.txt" % name), "w").write("\n".join(out))
    return "\n".join(out)


@pytest.mark.parametrize("name", SETS)
def test_cli_sam_identical(golden_unpacked, name):
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    cmd = [CLI, "-ax", "sr"]
    if m.get("rg"):
        cmd += ["-R", m["rg"]]
    r = subprocess.run(cmd + [m["ref"]] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert hashlib.md5(exp).hexdigest() == m["sam_md5"]
    assert r.stdout == exp, _diff_report(r.stdout, exp, name)


@pytest.mark.parametrize("name", ["g2_250pe", "g3_adversarial"])
def test_lds_dp_path_identical(golden_unpacked, name):
    """AL_DBG=128 forces every extension through the LDS-row DP (the path long targets take) instead of the
    register-resident one: both must give the reference's bytes."""
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    cmd = [CLI, "-ax", "sr"] + (["-R", m["rg"]] if m.get("rg") else [])
    r = subprocess.run(cmd + [m["ref"]] + m["reads"], cwd=d, capture_output=True, env=dict(os.environ, AL_DBG="128"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert r.stdout == exp, _diff_report(r.stdout, exp, name + "_lds")


def test_yeast100k_digest(tmp_path):
    """G5: 100 k pairs on the 12 Mbp synthetic genome (the bench workload's shape): md5 of the whole SAM must equal the
    digest the reference build produced (tests/golden/g5_yeast100k/meta.json)."""
    import gen_synth
    from conftest import GOLDEN
    m = json.load(open(os.path.join(GOLDEN, "g5_yeast100k", "meta.json")))
    gen_synth.generate(m["config"], str(tmp_path), pairs=m["pairs"], seed=m["seed"])
    r = subprocess.run([CLI, "-ax", "sr", "ref.fa", "reads_1.fq", "reads_2.fq"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert hashlib.md5(r.stdout).hexdigest() == m["sam_md5"]


def test_bwa_style_argv(golden_unpacked):
    """B1: `<aligner> mem -R RG -t N REF R1 R2` (src/0-align_reads.sh:13) gives the same records."""
    d = golden_unpacked["g1_mt150pe"]
    m = json.load(open(os.path.join(d, "meta.json")))
    r = subprocess.run([CLI, "mem", "-R", m["rg"], "-t", "4", m["ref"]] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == open(os.path.join(d, "expected.sam"), "rb").read()


def test_long_reads_fail_loudly(golden_unpacked):
    """Reads beyond the device extension kernel's limit must produce an error, never a silent fallback."""
    d = golden_unpacked["g4_MT_orang"]
    m = json.load(open(os.path.join(d, "meta.json")))
    r = subprocess.run([CLI, "-ax", "sr", m["ref"]] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode != 0
    assert b"not supported" in r.stderr


def test_map_frag_api(golden_unpacked):
    """al_map_frag (minimap.h:334 analogue) on single fragments equals the batch results."""
    import airlift_amd as A
    import ctypes as C
    from gpu_util import load_fragments
    d = golden_unpacked["g3_adversarial"]
    m, n_segs, seqs, names, quals = load_fragments(d)
    idx = A.Index(fasta=os.path.join(d, m["ref"]))
    ctx = A.Context(idx)
    ctx.upload(n_segs[:50], seqs[:100], names[:100]); ctx.run()
    nb, rb, _ = ctx.fetch()
    L = A.load()
    for f in range(0, 50, 7):
        ql = (C.c_int * 2)(len(seqs[2 * f]), len(seqs[2 * f + 1]))
        sq = (C.c_char_p * 2)(seqs[2 * f], seqs[2 * f + 1])
        nr = (C.c_int * 2)(); rg = (C.POINTER(A.Reg) * 2)()
        L.al_map_frag(idx.h, 2, ql, sq, nr, rg, ctx.h, C.byref(idx.mo), names[2 * f])
        for s in range(2):
            assert nr[s] == nb[2 * f + s]
            for k in range(nr[s]):
                a, b = rg[s][k], rb[2 * f + s][k]
                assert (a.rid, a.rs, a.re, a.qs, a.qe, a.mapq, a.rev, a.n_cigar, a.dp_max) == (b.rid, b.rs, b.re, b.qs, b.qe, b.mapq, b.rev, b.n_cigar, b.dp_max)
    ctx.close(); idx.close()
