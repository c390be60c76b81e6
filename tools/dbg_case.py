#!/usr/bin/env python3
"""Debug helper: maps a kept fuzz case (dir with ref.fa, r_1.fq[, r_2.fq]) through the library of the tree given as argv[2] and
dumps anchors / chains of one fragment (argv[3]) to argv[4].  usage: dbg_case.py CASE_DIR TREE FRAG OUT.npz [opt=val ...]"""
import os, sys
import numpy as np
d, tree, frag, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, tree)
import airlift_amd as A
idx = A.Index(fasta=os.path.join(d, "ref.fa"))
for kv in sys.argv[5:]:
    k, v = kv.split("="); setattr(idx.mo, k, type(getattr(idx.mo, k))(float(v)) if isinstance(getattr(idx.mo, k), float) else int(v))
n1, s1, _ = A.read_fastx(os.path.join(d, "r_1.fq"))
pe = os.path.exists(os.path.join(d, "r_2.fq"))
if pe:
    n2, s2, _ = A.read_fastx(os.path.join(d, "r_2.fq"))
    seqs = [x for p in zip(s1, s2) for x in p]; names = [x for p in zip(n1, n2) for x in p]; n_segs = [2] * len(s1)
else:
    seqs, names, n_segs = s1, n1, [1] * len(s1)
ctx = A.Context(idx)
ctx.upload(n_segs, seqs, names); ctx.run()
nf = len(n_segs)
na = ctx.tap("frag_na", np.uint32, nf); off = ctx.tap("a_off", np.uint64, nf + 1); nu = ctx.tap("frag_nu", np.uint32, nf)
tot = int(off[-1])
anchors = ctx.tap("anchors", np.uint64, tot * 2).reshape(-1, 2); chained = ctx.tap("chained", np.uint64, tot * 2).reshape(-1, 2)
u = ctx.tap("u", np.uint64, tot + nf + 1)
o = int(off[frag]); n = int(na[frag]); k = int(nu[frag])
uu = u[o + frag:o + frag + k]
nc = int((uu & np.uint64(0xffffffff)).sum())
np.savez(out, na=n, nu=k, anchors=anchors[o:o + n], u=uu, chained=chained[o:o + nc], all_na=na, all_nu=nu, all_off=off, all_anchors=anchors, all_u=u, all_chained=chained)
print(tree, "frag", frag, "na", n, "nu", k, "chained", nc)
