"""-m gpu tests of SURVEY.md N4 (al_prefilter.hip): the pre-alignment filters of the bundled mrFAST fork on the candidate locations the
minimap2 fork counts (ALSER loop, map.c:299-312).
  * GreedySnake: every decision of the kernel against the reference's own GreedySnake() -- GreedySnake.c:52 compiled where it lies into
    oracle/_ref/libgreedysnake.so (oracle/Makefile) -- on the same read and reference window;
  * adjacency filter (MrFAST.c:1741-1764 on minimizer seeds): against a restatement in this file over the oracle's sketch of the
    reference and of the reads (test infrastructure).
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
    """(candidates, kept by adjacency, kept by GreedySnake, kept by both) for the single-end reads of golden set d, and the device's four."""
    import airlift_amd
    m, n_segs, seqs, names, quals = load_fragments(d)
    assert set(n_segs) == {1}
    idx = A.Index(fasta=os.path.join(d, m["ref"]))
    ctx = A.Context(idx)
    ctx.upload(n_segs, seqs, names)
    L = A.load()
    L.al_batch_prefilter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]; L.al_batch_prefilter.restype = C.c_int
    got = (C.c_int64 * 4)()
    assert L.al_batch_prefilter(ctx.h, adj_e, snake_e, snake_k, snake_iter, got) == 0
    nf = len(seqs)
    na = ctx.tap("frag_na", np.uint32, nf); off = ctx.tap("a_off", np.uint64, nf + 1)
    anchors = ctx.tap("anchors", np.uint64, int(off[nf]) * 2).reshape(-1, 2)
    # reference: sequences as codes, minimizer -> sorted occurrence words (rid << 32 | pos << 1 | strand), as mm_idx_get returns them
    orc = OracleLib(); snake = _snake_lib()
    ref = airlift_amd.read_fastx(os.path.join(d, m["ref"]))
    ref_codes = [NT4[np.frombuffer(s, dtype=np.uint8)] for s in ref[1]]
    occ = {}
    for rid, s in enumerate(ref[1]):
        mz = orc.sketch(s, w, k)
        for x, y in mz:
            occ.setdefault(int(x) >> 8, []).append(rid << 32 | (int(y) & 0xffffffff))
    for v in occ.values():
        v.sort()
    exp = [0, 0, 0, 0]
    for f in range(nf):
        s = seqs[f]; Lr = len(s)
        fw = NT4[np.frombuffer(s, dtype=np.uint8)]; rv = np.where(fw[::-1] < 4, 3 - fw[::-1], 4).astype(np.uint8)
        seeds = []                                                  # collect_matches (map.c:90-123): query minimizers with 0 < occurrences < mid_occ
        for x, y in orc.sketch(s, w, k):
            lst = occ.get(int(x) >> 8)
            if lst and len(lst) < mid_occ:
                seeds.append((int(y) & 0xffffffff, lst))
        a = anchors[int(off[f]): int(off[f]) + int(na[f])]
        xs = (a[:, 0] & np.uint64(0xffffffff)).astype(np.int64); xs = np.where(xs >= 2**31, xs - 2**32, xs)
        seed_num, cs = 0, 0
        for i in range(1, len(a)):
            if xs[i] - xs[i - 1] > Lr:                              # map.c:301-308
                if seed_num >= min_cnt - 1:
                    exp[0] += 1
                    ax, ay = int(a[cs, 0]), int(a[cs, 1])
                    rev = ax >> 63; rid = (ax << 1 & (2**64 - 1)) >> 33
                    rpos = ax & 0xffffffff; qpos = ay & 0xffffffff
                    ref_start = rpos - qpos
                    diff = 0                                        # MrFAST.c:1741-1764
                    for qp, lst in seeds:
                        qend, qs = qp >> 1, qp & 1
                        rp = ref_start + (Lr - (qend + 1 - k) - 1) if rev else ref_start + qend
                        word = rid << 32 | rp << 1 | ((1 - qs) if rev else qs)
                        hit = 0 <= rp < len(ref_codes[rid]) and word in lst
                        if not hit:
                            diff += 1
                    keep_adj = diff <= adj_e
                    win = np.full(Lr, 5, dtype=np.uint8)
                    lo, hi = max(0, ref_start), min(len(ref_codes[rid]), ref_start + Lr)
                    if hi > lo:
                        win[lo - ref_start: hi - ref_start] = ref_codes[rid][lo:hi]
                    rd = rv if rev else fw
                    keep_snk = snake.GreedySnake(Lr, (win + 48).tobytes(), (rd + 48).tobytes(), snake_e, snake_k, 0, snake_iter) != 0
                    exp[1] += keep_adj; exp[2] += keep_snk; exp[3] += keep_adj and keep_snk
                seed_num, cs = 0, i
            else:
                seed_num += 1
    ctx.close(); idx.close()
    return exp, [int(v) for v in got]


@pytest.fixture(scope="module")
def A():
    import airlift_amd
    airlift_amd.load()
    return airlift_amd


@pytest.mark.parametrize("par", [(3, 3, 5, 3), (0, 1, 10, 1), (8, 6, 4, 5), (1, 0, 7, 2)], ids=["e3_k5", "e0_e1_k10", "e8_e6_k4", "e1_e0_k7"])
def test_filters_match_the_reference_functions(A, oracle_bin, golden_unpacked, par):
    exp, got = _expected(A, golden_unpacked["g2_100se"], *par)
    assert exp[0] > 100, "too few candidates for a meaningful comparison: %s" % exp
    assert got == exp


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
