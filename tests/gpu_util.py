"""Helpers shared by the -m gpu parity tests (product via the C-ABI vs oracle / goldens)."""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMP = bytes.maketrans(b"ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", b"TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn")


def revcomp(s):
    return s.translate(COMP)[::-1]


def qname_len(n):
    return len(n) - 2 if len(n) >= 3 and n[-1:].isdigit() and n[-2:-1] == b"/" else len(n)


def load_fragments(d):
    """Returns (n_segs, seqs, names, quals) in the reference's fragment order for golden set dir d."""
    import airlift_amd as A
    m = json.load(open(os.path.join(d, "meta.json")))
    files = [A.read_fastx(os.path.join(d, r)) for r in m["reads"]]
    n_segs, seqs, names, quals = [], [], [], []
    if len(files) == 2:
        (n1, s1, q1), (n2, s2, q2) = files
        for i in range(min(len(s1), len(s2))):
            n_segs.append(2); seqs += [s1[i], s2[i]]; names += [n1[i], n2[i]]; quals += [q1[i], q2[i]]
    else:
        nm, sq, ql = files[0]
        i = 0
        while i < len(sq):
            if i + 1 < len(sq) and nm[i][:qname_len(nm[i])] == nm[i + 1][:qname_len(nm[i + 1])]:
                n_segs.append(2); seqs += sq[i:i + 2]; names += nm[i:i + 2]; quals += ql[i:i + 2]; i += 2
            else:
                n_segs.append(1); seqs.append(sq[i]); names.append(nm[i]); quals.append(ql[i]); i += 1
    return m, n_segs, seqs, names, quals


class OracleLib:
    """ctypes view of oracle/libal_oracle.so (test infrastructure)."""

    class O128(C.Structure):
        _fields_ = [("x", C.c_uint64), ("y", C.c_uint64)]

    def __init__(self):
        self.L = C.CDLL(os.path.join(ROOT, "oracle", "libal_oracle.so"))
        self.L.o_sketch.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.POINTER(self.O128)), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.L.o_sketch.restype = None
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    def sketch(self, seq, w=11, k=21):
        if len(seq) == 0:
            return np.zeros((0, 2), dtype=np.uint64)
        a = C.POINTER(self.O128)(); n = C.c_size_t(0); m = C.c_size_t(0)
        self.L.o_sketch(seq, len(seq), w, k, 0, C.byref(a), C.byref(n), C.byref(m))
        out = np.array([(a[i].x, a[i].y) for i in range(n.value)], dtype=np.uint64).reshape(-1, 2)
        if n.value:
            self.libc.free(a)
        return out


def seed_lines(tag, names, xy, cid=None):
    """Formats anchors like the reference's --print-seeds taps (map.c:333-338,381-385)."""
    out = []
    x = xy[:, 0]; y = xy[:, 1]
    xi = (x & np.uint64(0xffffffff)).astype(np.int64); xi = np.where(xi >= 2**31, xi - 2**32, xi)
    yi = (y & np.uint64(0xffffffff)).astype(np.int64); yi = np.where(yi >= 2**31, yi - 2**32, yi)
    rid = ((x << np.uint64(1)) >> np.uint64(33)).astype(np.int64)
    rev = (x >> np.uint64(63)).astype(np.int64)
    span = ((y >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
    for i in range(len(x)):
        gap = 0 if i == 0 else (yi[i] - yi[i - 1]) - (xi[i] - xi[i - 1])
        pre = "%s\t%d\t" % (tag, cid) if cid is not None else "%s\t" % tag
        out.append("%s%s\t%d\t%s\t%d\t%d\t%d" % (pre, names[rid[i]], xi[i], "+-"[rev[i]], yi[i], span[i], gap))
    return out


def split_expected_seeds(text):
    """Splits the golden --seeds tap into per-fragment blocks (each starts with an RS line)."""
    blocks, cur = [], None
    for l in text.split("\n"):
        if not l:
            continue
        if l.startswith("RS\t"):
            if cur is not None:
                blocks.append(cur)
            cur = [l]
        else:
            cur.append(l)
    if cur is not None:
        blocks.append(cur)
    return blocks
