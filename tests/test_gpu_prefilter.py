"""-m gpu tests of SURVEY.md N4 (al_prefilter.hip): the pre-alignment filters of the bundled mrFAST fork on the candidate locations the
minimap2 fork counts (ALSER loop, map.c:299-312).
  * GreedySnake: every decision of the kernel against the reference's own GreedySnake() -- GreedySnake.c:52 compiled where it lies into
    oracle/_ref/libgreedysnake.so (oracle/Makefile) -- on the same read and reference window;
  * adjacency filter (MrFAST.c:1741-1764 on minimizer seeds): against oracle/n4_oracle.py over the oracle's sketch of the reference and of the
    reads;
  * candidate by candidate (al_batch_prefilter_decisions), not by aggregate counts.
The candidates are rebuilt here from the device's sorted anchors, which tests/test_gpu_stages.py pins to the reference's --print-seeds taps."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from gpu_util import OracleLib, load_fragments

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
NT4 = np.full(256, 4, dtype=np.uint8)
for i, ch in enumerate(b"ACGT"):
    NT4[ch] = i; NT4[ch + 32] = i
NT4[ord("U")] = NT4[ord("u")] = 3


def _snake_lib():
    p = os.path.join(ROOT, "oracle", "_ref", "libgreedysnake.so")
    if not os.path.exists(p):
        pytest.fail("oracle/_ref/libgreedysnake.so is missing: the reference's GreedySnake.c must be built by oracle/Makefile (target ref) and travel with the snapshot")
    L = C.CDLL(p)
    L.GreedySnake.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]; L.GreedySnake.restype = C.c_int
    return L


def _expected(A, d, adj_e, snake_e, snake_k, snake_iter, k=21, w=11, mid_occ=1000, min_cnt=2):
    """Per candidate of golden set d (every read as a single-segment fragment): {(fragment, first anchor of its cluster): (kept by adjacency,
    kept by GreedySnake)} by the oracle / the reference's GreedySnake, and the device's dictionary and four counts."""
    import sys
    import airlift_amd
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import n4_oracle
    m, n_segs, seqs, names, quals = load_fragments(d)
    seqs = [s for s in seqs if 0 < len(s) <= 512]
    idx = A.Index(fasta=os.path.join(d, m["ref"]))
    ctx = A.Context(idx)
    ctx.upload([1] * len(seqs), seqs, [b""] * len(seqs))
    L = A.load()
    L.al_batch_prefilter_decisions.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_int64]; L.al_batch_prefilter_decisions.restype = C.c_int
    got = (C.c_int64 * 4)()
    cap = 1 << 20; dec = np.zeros(cap, dtype=np.uint64)
    assert L.al_batch_prefilter_decisions(ctx.h, adj_e, snake_e, snake_k, snake_iter, got, dec.ctypes.data_as(C.c_void_p), cap) == 0
    assert got[0] <= cap
    got_dec = {(int(v) >> 32, (int(v) & 0xffffffff) >> 2): (bool(int(v) & 2), bool(int(v) & 1)) for v in dec[: got[0]]}
    assert len(got_dec) == got[0], "a candidate was reported twice"
    nf = len(seqs)
    na = ctx.tap("frag_na", np.uint32, nf); off = ctx.tap("a_off", np.uint64, nf + 1)
    anchors = ctx.tap("anchors", np.uint64, int(off[nf]) * 2).reshape(-1, 2)
    # reference: sequences as codes, minimizer -> sorted occurrence words (rid << 32 | pos << 1 | strand), as mm_idx_get returns them
    orc = OracleLib(); snake = _snake_lib()
    ref = airlift_amd.read_fastx(os.path.join(d, m["ref"]))
    ref_codes = [NT4[np.frombuffer(s, dtype=np.uint8)] for s in ref[1]]
    occ = {}
    for rid, s in enumerate(ref[1]):
        for x, y in orc.sketch(s, w, k):
            occ.setdefault(int(x) >> 8, []).append(rid << 32 | (int(y) & 0xffffffff))
    occ = {h: (sorted(v), set(v)) for h, v in occ.items()}
    exp = {}
    for f in range(nf):
        s = seqs[f]; Lr = len(s)
        fw = NT4[np.frombuffer(s, dtype=np.uint8)]; rv = np.where(fw[::-1] < 4, 3 - fw[::-1], 4).astype(np.uint8)
        seeds = []                                                  # collect_matches (map.c:90-123): query minimizers with 0 < occurrences < mid_occ
        for x, y in orc.sketch(s, w, k):
            e = occ.get(int(x) >> 8)
            if e and len(e[0]) < mid_occ:
                seeds.append((int(y) & 0xffffffff, e[1]))
        a = anchors[int(off[f]): int(off[f]) + int(na[f])]
        for cs in n4_oracle.candidates(a, Lr, min_cnt):
            rev, rid, ref_start = n4_oracle.candidate_location(a[cs, 0], a[cs, 1])
            keep_adj = n4_oracle.adjacency(seeds, rev, rid, ref_start, Lr, k, len(ref_codes[rid]), adj_e)
            win = np.full(Lr, 5, dtype=np.uint8)
            lo, hi = max(0, ref_start), min(len(ref_codes[rid]), ref_start + Lr)
            if hi > lo:
                win[lo - ref_start: hi - ref_start] = ref_codes[rid][lo:hi]
            rd = rv if rev else fw
            keep_snk = snake.GreedySnake(Lr, (win + 48).tobytes(), (rd + 48).tobytes(), snake_e, snake_k, 0, snake_iter) != 0     # the reference's own function (GreedySnake.c:52)
            exp[(f, cs)] = (bool(keep_adj), bool(keep_snk))
    ctx.close(); idx.close()
    return exp, got_dec, [int(v) for v in got]


@pytest.fixture(scope="module")
def A():
    import airlift_amd
    airlift_amd.load()
    return airlift_amd


@pytest.mark.parametrize("name", ["g2_100se", "g1_mt150pe", "g3_adversarial", "g6_repeats"])
@pytest.mark.parametrize("par", [(3, 3, 5, 3), (0, 1, 10, 1), (8, 6, 4, 5), (1, 0, 7, 2)], ids=["e3_k5", "e0_e1_k10", "e8_e6_k4", "e1_e0_k7"])
def test_filters_match_the_reference_functions(A, oracle_bin, golden_unpacked, name, par):
    """EVERY candidate's two decisions: GreedySnake against the reference's own function on the same read and window, the adjacency filter against
    oracle/n4_oracle.py; on four golden sets (their reads taken as single-end) and four parameter sets.  The four counts follow."""
    exp, got_dec, got = _expected(A, golden_unpacked[name], *par)
    if name == "g2_100se":
        assert len(exp) > 100, "too few candidates for a meaningful comparison: %d" % len(exp)
    assert set(got_dec) == set(exp), "candidate sets differ: %s" % sorted(set(got_dec) ^ set(exp))[:5]
    bad = [(c, got_dec[c], exp[c]) for c in exp if got_dec[c] != exp[c]]
    assert not bad, "%d of %d decisions differ, first: %s" % (len(bad), len(exp), bad[:5])
    assert got == [len(exp), sum(v[0] for v in exp.values()), sum(v[1] for v in exp.values()), sum(v[0] and v[1] for v in exp.values())]


def test_cli_reports_the_filtered_counts(golden_unpacked):
    """--count-candidates --prefilter E,SE,K,I: the first line is the fork's count (pinned by test_gpu_sam / the mm2count golden), the
    three lines behind it are this file's numbers."""
    d = golden_unpacked["g2_100se"]
    m = json.load(open(os.path.join(d, "meta.json")))
    plain = subprocess.run([CLI, "-ax", "sr", "--count-candidates", m["ref"], m["reads"][0]], cwd=d, capture_output=True)
    filt = subprocess.run([CLI, "-ax", "sr", "--count-candidates", "--prefilter", "3,3,5,3", m["ref"], m["reads"][0]], cwd=d, capture_output=True)
    assert plain.returncode == 0 and filt.returncode == 0, filt.stderr.decode()[-1000:]
    line = [l for l in plain.stderr.decode().split("\n") if l.startswith("Total No. of Mappings")]
    assert line and line[0] in filt.stderr.decode()
    assert "Candidates kept by the adjacency filter" in filt.stderr.decode() and "Candidates kept by GreedySnake" in filt.stderr.decode()
