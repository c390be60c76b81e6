#!/usr/bin/env python3
"""Run one batch several times and report the first pipeline stage whose tapped output changes between runs."""
import sys, os, hashlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from airlift_amd import capi as A

d = sys.argv[1]; runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
r1 = A.read_fastx(os.path.join(d, "r_1.fq")); r2 = A.read_fastx(os.path.join(d, "r_2.fq"))
n_segs, seqs, names = [], [], []
for n1, s1, n2, s2 in zip(r1[0], r1[1], r2[0], r2[1]):
    n_segs.append(2); seqs += [s1, s2]; names += [n1, n2]
nf = len(n_segs)
idx = A.Index(fasta=os.path.join(d, "ref.fa"))
ctx = A.Context(idx)
base = None
for it in range(runs):
    ctx.upload(n_segs, seqs, names); ctx.run()
    st = ctx.stat(); tot = int(st.n_anchor)
    t = {}
    t["frag_na_p1"] = ctx.tap("frag_na_p1", np.uint32, nf); t["a_off_p1"] = ctx.tap("a_off_p1", np.uint64, nf + 1); t["frag_rep_p1"] = ctx.tap("frag_rep_p1", np.int32, nf)
    t["frag_na"] = ctx.tap("frag_na", np.uint32, nf); t["a_off"] = ctx.tap("a_off", np.uint64, nf + 1); t["frag_rep"] = ctx.tap("frag_rep", np.int32, nf)
    t["frag_nu"] = ctx.tap("frag_nu", np.uint32, nf)
    anchors = ctx.tap("anchors", np.uint64, tot * 2).reshape(-1, 2)
    chained = ctx.tap("chained", np.uint64, tot * 2).reshape(-1, 2)
    u = ctx.tap("u", np.uint64, tot + nf + 1)
    # per-fragment digests (layout may legitimately differ between runs)
    per = {"anchors_p1": [], "anchors": [], "u": [], "chained": []}
    for f in range(nf):
        o1, n1 = int(t["a_off_p1"][f]), int(t["frag_na_p1"][f]); o, n = int(t["a_off"][f]), int(t["frag_na"][f]); nu = int(t["frag_nu"][f])
        per["anchors_p1"].append(hashlib.md5(anchors[o1:o1 + n1].tobytes()).hexdigest())
        per["anchors"].append(hashlib.md5(anchors[o:o + n].tobytes()).hexdigest())
        uu = u[o + f:o + f + nu]
        per["u"].append(hashlib.md5(uu.tobytes()).hexdigest())
        nc = int((uu & 0xffffffff).sum())
        per["chained"].append(hashlib.md5(chained[o:o + nc].tobytes()).hexdigest())
    n_regs, regs, rep = ctx.fetch()
    rg = []
    for r in range(len(seqs)):
        rg.append(hashlib.md5(b"".join(bytes(regs[r][i]) for i in range(n_regs[r]))).hexdigest())
    raw_u = {f: u[int(t["a_off"][f]) + f: int(t["a_off"][f]) + f + int(t["frag_nu"][f])].copy() for f in range(nf) if t["frag_na"][f] < 4000}
    raw_c = {f: chained[int(t["a_off"][f]): int(t["a_off"][f]) + int(t["frag_na"][f])].copy() for f in raw_u}
    raw_a = {f: anchors[int(t["a_off"][f]): int(t["a_off"][f]) + int(t["frag_na"][f])].copy() for f in raw_u}
    cur = dict(t); cur["raw_u"] = raw_u; cur["raw_c"] = raw_c; cur.update({k: np.array(v) for k, v in per.items()}); cur["regs"] = np.array(rg)
    if base is None: base = cur; print("run 0: anchors %d rechain %d" % (tot, st.n_rechain)); continue
    for k in ["frag_na_p1", "frag_rep_p1", "anchors_p1", "frag_na", "frag_rep", "anchors", "frag_nu", "u", "chained", "regs"]:
        bad = np.nonzero(base[k] != cur[k])[0]
        if len(bad):
            f = int(bad[0]); ff = f // 2 if k == "regs" else f
            print("run %d: %s differs at %d entries, first %d (%s): na_p1=%d na=%d nu=%d/%d rep=%d" % (it, k, len(bad), f, names[2 * ff], base["frag_na_p1"][ff], base["frag_na"][ff], base["frag_nu"][ff], cur["frag_nu"][ff], base["frag_rep"][ff]))
            if k == "u" and f in base["raw_u"]:
                fmt = lambda uu: " ".join("%d:%d" % (x >> 32, x & 0xffffffff) for x in uu)
                print("   base u: " + fmt(base["raw_u"][f])); print("   cur  u: " + fmt(cur["raw_u"][f]))
                k0 = 0
                for ci, (ub, uc) in enumerate(zip(base["raw_u"][f], cur["raw_u"][f])):
                    nb, nc = int(ub & 0xffffffff), int(uc & 0xffffffff)
                    cb, cc = base["raw_c"][f][k0:k0 + nb], cur["raw_c"][f][k0:k0 + nc]
                    if ub != uc or not np.array_equal(cb, cc):
                        print("   chain %d differs; base first/last x,y: %x %x .. %x %x ; cur: %x %x .. %x %x" % (ci, cb[0][0], cb[0][1], cb[-1][0], cb[-1][1], cc[0][0], cc[0][1], cc[-1][0], cc[-1][1]))
                        np.save("gpurun_out/nd_anchors.npy", raw_a[f]); np.save("gpurun_out/nd_base_c.npy", base["raw_c"][f]); np.save("gpurun_out/nd_cur_c.npy", cur["raw_c"][f])
                        np.save("gpurun_out/nd_base_u.npy", base["raw_u"][f]); np.save("gpurun_out/nd_cur_u.npy", cur["raw_u"][f])
                        break
                    k0 += nb
            break
    else:
        print("run %d: identical" % it)

# layout audit: every fragment's anchor region must end before the next one starts
off = base["a_off"][:nf].astype(np.int64); na = base["frag_na"].astype(np.int64)
o = np.argsort(off, kind="stable")
end = off[o] + na[o]
bad = np.nonzero(end[:-1] > off[o][1:])[0]
print("layout: %d overlapping neighbours" % len(bad))
for b in bad[:5]:
    print("   frag %d [%d,+%d) overlaps frag %d at %d" % (o[b], off[o[b]], na[o[b]], o[b + 1], off[o[b + 1]]))
print("   total anchors %d" % tot)
