"""-m gpu parity at the sizes of BASELINE.json's configs 3, 4 and 5 (SURVEY.md 8(d) workloads C3 / C4 / C5, generated here by
tools/gen_synth.py): the drop-in maps a sample of pairs against the full-size reference and its SAM must be byte-identical to
what the reference build (oracle/_ref/mm2ref, which travels with the snapshot) prints for the same files on this box.
Covers what the small golden sets cannot: the 2^30-slot table and > 2^32 base offsets of a 3.1 Gbp index, minimizers above
mid_occ / max_occ, fragments with thousands of anchors and hundreds of chains (segment chaining, block / device-wide anchor
sorts, the re-chain pass), 250 bp reads on the human-sized reference."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
REF = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def _first_diff(a, b):
    with open(a, "rb") as fa, open(b, "rb") as fb:
        for i, (la, lb) in enumerate(zip(fa, fb)):
            if la != lb:
                return "line %d\n  got %s\n  exp %s" % (i, la[:300], lb[:300])
    return "length differs"


class _Workload:
    """Reference FASTA of a config (generated once per session) + the CPU reference's index dump for it."""

    def __init__(self, config, base):
        import gen_synth as g
        self.g, self.config = g, config
        self.dir = str(base)
        self.ref = g.build_reference(config, cache_dir=os.environ.get("AL_REF_CACHE"))
        self.fa = os.path.join(self.dir, "ref.fa")
        g.write_fasta(self.fa, self.ref)
        self.mmi = None

    def reads(self, tag, pairs, read_len, seed):
        r1, r2 = self.g.simulate(self.config, self.ref, pairs, seed, read_len=read_len)
        p1, p2 = os.path.join(self.dir, tag + "_1.fq"), os.path.join(self.dir, tag + "_2.fq")
        self.g.write_fastq(p1, r1); self.g.write_fastq(p2, r2)
        return p1, p2

    def check(self, tag, p1, p2):
        if not os.path.exists(REF):
            pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot -- without it this comparison against the reference would silently not run")
        nt = str(min(os.cpu_count() or 1, 64))
        cpu, gpu = os.path.join(self.dir, tag + "_cpu.sam"), os.path.join(self.dir, tag + "_gpu.sam")
        with open(cpu, "wb") as f:      # first use builds the CPU index and saves it; later cases load the dump
            if self.mmi is None:
                mmi = os.path.join(self.dir, "ref.mmi")
                subprocess.run([REF, "-t", nt, "--save-index", mmi, self.fa, p1, p2], stdout=f, stderr=subprocess.DEVNULL, check=True)
                self.mmi = mmi
            else:
                subprocess.run([REF, "-t", nt, self.mmi, p1, p2], stdout=f, stderr=subprocess.DEVNULL, check=True)
        with open(gpu, "wb") as f:
            r = subprocess.run([CLI, "-ax", "sr", "-t", "16", self.fa, p1, p2], stdout=f, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        n = sum(1 for _ in open(gpu, "rb"))
        assert n > 100
        assert _md5(gpu) == _md5(cpu), _first_diff(gpu, cpu)
        for p in (cpu, gpu, p1, p2):
            os.remove(p)


@pytest.fixture(scope="module")
def c3(tmp_path_factory):
    return _Workload("c3", tmp_path_factory.mktemp("c3"))


@pytest.fixture(scope="module")
def c4(tmp_path_factory):
    w = _Workload("c4", tmp_path_factory.mktemp("c4"))
    yield w
    for p in (w.fa, w.mmi):
        if p and os.path.exists(p):
            os.remove(p)


def test_c3_150bp_200k_pairs(c3):
    """BASELINE config 3 (ce11-sized, 100.3 Mbp): 200 k mason-like 150 bp pairs."""
    p1, p2 = c3.reads("c3", 200_000, 150, seed=31)
    c3.check("c3", p1, p2)


def test_c4_150bp_200k_pairs(c4):
    """BASELINE config 4 (GRCh38-sized, 3.1 Gbp, 45 % repeats, 5 % N): 200 k mason-like 150 bp pairs."""
    p1, p2 = c4.reads("c4", 200_000, 150, seed=41)
    c4.check("c4", p1, p2)


def test_c5_250bp_100k_pairs(c4):
    """BASELINE config 5: 250 bp pairs, insert N(550, 60), on the same 3.1 Gbp reference."""
    p1, p2 = c4.reads("c5", 100_000, 250, seed=51)
    c4.check("c5", p1, p2)
