#!/usr/bin/env python3
"""Does running the batch as S independent sub-batches on S contexts (streams) of one GPU overlap latency-bound and
issue-bound kernels?  tools/streams_probe.py [pairs] [S ...]"""
import ctypes as C, os, sys, threading, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, gen_synth as g, airlift_amd as A
import bench
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
Ss = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
L = A.load()
L.al_batch_upload_flat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int64]
rk, _ = g.CONFIGS["c2"]; ref = g.make_reference(**rk)
tmp = tempfile.mkdtemp(); g.write_fasta(tmp + "/ref.fa", ref)
idx = A.Index(fasta=tmp + "/ref.fa", on_device=0)
arr = bench.make_workload("c2", pairs, 150, 20261002, ref)
for S in Ss:
    nf = pairs // S; ctxs = []
    for s in range(S):
        ctx = A.Context(idx, device=0); L.al_ctx_set_threads(ctx.h, 16)
        sub = np.ascontiguousarray(arr[s * nf:(s + 1) * nf])
        n_segs = (C.c_int * nf)(*([2] * nf)); qlens = (C.c_int * (2 * nf))(*([150] * (2 * nf)))
        assert L.al_batch_upload_flat(ctx.h, nf, n_segs, qlens, sub.ctypes.data_as(C.c_char_p), b"realigned_", s * nf) == 0
        ctx.n_frag, ctx.n_reads = nf, 2 * nf; ctxs.append(ctx)
    def run_all():
        th = [threading.Thread(target=c.run) for c in ctxs]
        [t.start() for t in th]; [t.join() for t in th]
    run_all(); run_all()
    t0 = time.perf_counter()
    for _ in range(4): run_all()
    dt = (time.perf_counter() - t0) / 4
    print("S=%d: %.1f ms per %d pairs -> %.1f M reads/s" % (S, dt * 1e3, nf * S, 2 * nf * S / dt / 1e6))
    for c in ctxs: c.close()
