#!/usr/bin/env python3
"""Experiment: resident throughput with N mapping contexts in flight (one host thread each) against one context.
usage: r6_two_ctx.py [pairs] [steps]"""
import ctypes as C, os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import bench as B
import gen_synth as g
import airlift_amd as A
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = A.load()
L.al_batch_upload_flat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int64]; L.al_batch_upload_flat.restype = C.c_int
L.al_ctx_set_no_taps.argtypes = [C.c_void_p, C.c_int]; L.al_ctx_set_no_taps.restype = None
ref = g.build_reference("c4"); tmp = tempfile.mkdtemp(prefix="al_two_"); fa = os.path.join(tmp, "ref.fa"); g.write_fasta(fa, ref)
idx = A.Index(fasta=fa, on_device=0)
def mk(lo):
    arr = B.make_workload("c4", lo, lo + pairs, 150, 20261002, ref, None)
    ctx = A.Context(idx, device=0); L.al_ctx_set_threads(ctx.h, 32); L.al_ctx_set_no_taps(ctx.h, 1)
    n_segs = (C.c_int * pairs)(*([2] * pairs)); qlens = (C.c_int * (2 * pairs))(*([150] * (2 * pairs)))
    assert L.al_batch_upload_flat(ctx.h, pairs, n_segs, qlens, arr.ctypes.data_as(C.c_char_p), b"realigned_", lo) == 0
    ctx.n_frag, ctx.n_reads = pairs, 2 * pairs
    return ctx
for nctx in (1, 2, 3):
    try:
        ctxs = [mk(i * pairs) for i in range(nctx)]
    except Exception as e:
        print("contexts", nctx, "failed:", e); break
    for c in ctxs: c.run()                                  # warm-up (workspaces)
    def work(c, n):
        for _ in range(n): c.run()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, steps)) for c in ctxs]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print("contexts %d x %d pairs: %d steps each in %.3f s -> %.2f ms per step of %d pairs, %.2f M reads/s" % (nctx, pairs, steps, dt, dt / (steps * nctx) * 1e3, pairs, 2 * pairs * steps * nctx / dt / 1e6), flush=True)
    for c in ctxs: c.close()
