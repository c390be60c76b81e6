import sys, os, numpy as np, ctypes as C, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import gen_synth as g
import airlift_amd as A
import bench
rk, _ = g.CONFIGS["c2r"]
ref = g.make_reference(**rk)
tmp = tempfile.mkdtemp(); g.write_fasta(os.path.join(tmp, "ref.fa"), ref)
idx = A.Index(fasta=os.path.join(tmp, "ref.fa"), on_device=0)
nf = 200000
arr = bench.make_workload(nf, 150, 20261002, ref, 0)
L = A.load()
L.al_batch_upload_flat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int64]
ctx = A.Context(idx, device=0)
n_segs = (C.c_int * nf)(*([2] * nf)); qlens = (C.c_int * (2 * nf))(*([150] * (2 * nf)))
L.al_batch_upload_flat(ctx.h, nf, n_segs, qlens, arr.ctypes.data_as(C.c_char_p), b"realigned_", 0)
ctx.n_frag, ctx.n_reads = nf, 2 * nf
ctx.run()
na = ctx.tap("frag_na", np.uint32, nf); nu = ctx.tap("frag_nu", np.uint32, nf)
print("anchors/frag percentiles", np.percentile(na, [50, 90, 95, 98, 99, 99.9, 100]))
print("frags n>128:", (na > 128).sum(), " n>768:", (na > 768).sum(), " sum anchors n>128:", na[na > 128].sum(), "of", na.sum())
print("chains/frag percentiles", np.percentile(nu, [50, 90, 99, 99.9, 100]), "frags nu>64:", (nu > 64).sum())
st = ctx.stat(); print("stage ms", [round(x, 2) for x in list(st.ms_kernel)[:22]])
