"""CPU test: the closed-form result the device uses for extension jobs whose query matches the target with at most
one mismatch (no N, query not longer than the target) must equal the reference DP (oracle o_ksw_extd2 = ksw_extd2_sse)
field by field, for both gap-alignment modes used by mm_align1 (left: RIGHT|REV_CIGAR|EXTZ_ONLY, right: EXTZ_ONLY)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Ez(C.Structure):
    _fields_ = [("max", C.c_uint32), ("zdropped", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int), ("mqe", C.c_int), ("mqe_t", C.c_int),
                ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int), ("reach_end", C.c_int),
                ("cigar", C.POINTER(C.c_uint32))]


def predicted(q, t, a=2, b=8, end_bonus=10, w=151):
    """Closed form ('diagonal shortcut'): returns None when the shortcut does not apply."""
    ql, tl = len(q), len(t)
    if ql == 0 or ql > tl or (q >= 4).any() or (t[:ql] >= 4).any():
        return None
    mm = q != t[:ql]
    if mm.sum() > 1:
        return None
    s = np.cumsum(np.where(mm, -b, a))
    mx, pos = 0, -1
    for j in range(ql):
        if s[j] > mx:
            mx, pos = int(s[j]), j
    runoff = tl >= ql + w + 1        # the band leaves the matrix at row 2*ql + w - 1: flagged like a z-drop (ksw2_extd2_sse.c:135)
    reach = (not runoff) and int(s[-1]) + end_bonus > mx
    ncig = 1 if (reach or pos >= 0) else 0
    cig = ((ql if reach else pos + 1) << 4) if ncig else None
    return dict(max=mx, max_q=pos, max_t=pos, mqe_t=ql - 1, reach_end=int(reach), n_cigar=ncig, cigar=cig, zdropped=int(runoff))


@pytest.fixture(scope="module")
def lib(oracle_bin):
    L = C.CDLL(os.path.join(ROOT, "oracle", "libal_oracle.so"))
    L.o_ksw_extd2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int8, C.POINTER(C.c_int8), C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Ez)]
    L.o_ksw_extd2.restype = None
    return L


def test_closed_form_equals_dp(lib):
    rng = np.random.default_rng(5)
    mat = (C.c_int8 * 25)()
    for i in range(5):
        for j in range(5):
            mat[i * 5 + j] = -1 if (i == 4 or j == 4) else (2 if i == j else -8)
    n_checked = 0
    for it in range(6000):
        w = [151, 151, 76, 31, 226][it % 5 if it % 7 else 0]
        ql = int(rng.integers(1, 90 if it % 3 else 260)); tl = ql + int(rng.integers(0, 100 if it % 4 else 300))
        kind = it % 5
        if kind == 0:      # homopolymer / dinucleotide repeats: shifted alignments score as well as they ever can
            unit = rng.integers(0, 4, size=int(rng.integers(1, 4)), dtype=np.uint8)
            t = np.tile(unit, tl // len(unit) + 2)[:tl].copy()
        else:
            t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        q = t[:ql].copy()
        nm = int(rng.integers(0, 2))
        for _ in range(nm):
            p = int(rng.integers(0, ql)); q[p] = (q[p] + int(rng.integers(1, 4))) & 3
        exp = predicted(q, t, w=w)
        if exp is None:
            continue
        for flag in (0x40, 0x40 | 0x02 | 0x80):
            ez = Ez()
            lib.o_ksw_extd2(ql, q.tobytes(), tl, t.tobytes(), 5, mat, 12, 2, 24, 1, w, 100, 10, flag, C.byref(ez))
            got = dict(max=ez.max, max_q=ez.max_q, max_t=ez.max_t, mqe_t=ez.mqe_t, reach_end=ez.reach_end, n_cigar=ez.n_cigar,
                       cigar=ez.cigar[0] if ez.n_cigar else None, zdropped=int(ez.zdropped != 0))
            # mqe_t is only consumed when reach_end is set (align.c:702,769)
            if not exp["reach_end"]:
                got["mqe_t"] = exp["mqe_t"]
            assert got == exp, (it, flag, ql, tl, nm, got, exp)
            n_checked += 1
    assert n_checked > 8000
