#!/usr/bin/env python3
"""How many mapping contexts does it take to saturate the GPU at the drop-in's batch size?  N host threads, each with its own context
holding the same resident batch of PAIRS C4 pairs, run K steps each; aggregate reads/s for N = 1, 2, 3, 4, 6.  (The mapping kernels alone:
no parsing, no SAM, no file I/O.)  usage: multi_ctx.py [pairs] [steps]"""
import ctypes as C, os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import gen_synth as g
import airlift_amd as A
import bench
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = A.load()
L.al_batch_upload_flat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int64]; L.al_batch_upload_flat.restype = C.c_int
ref = g.build_reference("c4"); tmp = tempfile.mkdtemp(); g.write_fasta(os.path.join(tmp, "ref.fa"), ref)
idx = A.Index(fasta=os.path.join(tmp, "ref.fa"), on_device=0)
arr = bench.make_workload("c4", 0, pairs, 150, 20261002, ref, None)
n_segs = (C.c_int * pairs)(*([2] * pairs)); qlens = (C.c_int * (2 * pairs))(*([150] * (2 * pairs)))
ctxs = []
for i in range(6):
    c = A.Context(idx, device=0)
    assert L.al_batch_upload_flat(c.h, pairs, n_segs, qlens, arr.ctypes.data_as(C.c_char_p), b"r_", 0) == 0
    c.n_frag, c.n_reads = pairs, 2 * pairs
    c.run(); ctxs.append(c)
for n in (1, 2, 3, 4, 6, 3, 1):
    def work(c):
        for _ in range(steps): L.al_batch_run(c.h)
    th = [threading.Thread(target=work, args=(ctxs[i],)) for i in range(n)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print("%d context(s) x %d steps of %d pairs: %.1f ms per step and context, %.2f M reads/s in all" % (n, steps, pairs, dt / steps * 1e3, n * steps * 2 * pairs / dt / 1e6), flush=True)
