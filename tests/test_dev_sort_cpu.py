"""CPU test: the device restatement of the reference's sort (al_dev_sort.h: stable insertion sort up to 64 elements, the
in-place MSD radix permutation of ksort.h:116-151 above) leaves equal keys in exactly the reference's order.  Checked
against the oracle (o_radix_sort_128x) and, when /root/reference is present, against the reference's ksort.h itself."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    rng = np.random.default_rng(11)
    out = []
    for it in range(400):
        n = int(rng.choice([1, 2, 63, 64, 65, 66, 100, 129, 300, 1000, 2500, 4200, 9000][it % 13:][:1] or [65]))
        kind = it % 6
        if kind == 0:    x = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)                                    # distinct
        elif kind == 1:  x = rng.integers(0, 4, size=n, dtype=np.uint64)                                          # a few keys: every bucket recursion sees ties
        elif kind == 2:  x = rng.integers(0, 50, size=n, dtype=np.uint64) << np.uint64(int(rng.integers(0, 57)))  # ties inside one byte position
        elif kind == 3:  x = (rng.integers(0, 3, size=n, dtype=np.uint64) << np.uint64(63)) | (rng.integers(0, 5, size=n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 40, size=n, dtype=np.uint64)  # anchor-like: strand | contig | position
        elif kind == 4:  x = np.full(n, 12345, dtype=np.uint64)                                                   # all equal
        else:            x = np.sort(rng.integers(0, 300, size=n, dtype=np.uint64))[::-1].copy()                  # descending with ties
        out.append(x)
    return out


@pytest.fixture(scope="module")
def host_sort(tmp_path_factory):
    d = tmp_path_factory.mktemp("sort")
    so = os.path.join(d, "libsort_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "airlift_amd", "csrc"), os.path.join(ROOT, "tests", "csrc", "sort_host.cpp"), "-o", so])
    L = C.CDLL(so)
    L.t_sort128.argtypes = [C.c_void_p, C.c_int]; L.t_sort_perm.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return L


def _pairs(x):
    a = np.empty((len(x), 2), dtype=np.uint64); a[:, 0] = x; a[:, 1] = np.arange(len(x), dtype=np.uint64)   # y = original index: ties are distinguishable
    return a


def test_matches_oracle_sort(host_sort, oracle_bin):
    O = C.CDLL(os.path.join(ROOT, "oracle", "libal_oracle.so"))
    O.o_radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
    for x in _cases():
        a, b = _pairs(x), _pairs(x)
        assert host_sort.t_sort128(a.ctypes.data, len(x)) == 0
        O.o_radix_sort_128x(b.ctypes.data, b.ctypes.data + b.nbytes)
        assert np.array_equal(a, b), "n=%d" % len(x)
        assert (np.diff(a[:, 0].astype(np.float64)) >= 0).all()
        # the permutation form used for ordering chains gives the same order of ids
        t = np.arange(len(x), dtype=np.int32); keys = np.ascontiguousarray(x)
        assert host_sort.t_sort_perm(t.ctypes.data, keys.ctypes.data, len(x)) == 0
        assert np.array_equal(t.astype(np.uint64), a[:, 1])


def test_matches_reference_ksort(host_sort, tmp_path):
    hdr = glob.glob("/root/reference/**/ksort.h", recursive=True)
    if not hdr:
        pytest.skip("reference sources not present (GPU box)")
    so = os.path.join(tmp_path, "libsort_ref.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", '-DREF_KSORT_H="%s"' % hdr[0], os.path.join(ROOT, "tests", "csrc", "sort_ref.c"), "-o", so])
    R = C.CDLL(so); R.ref_sort128.argtypes = [C.c_void_p, C.c_int]
    for x in _cases():
        a, b = _pairs(x), _pairs(x)
        host_sort.t_sort128(a.ctypes.data, len(x)); R.ref_sort128(b.ctypes.data, len(x))
        assert np.array_equal(a, b), "n=%d" % len(x)
