#!/usr/bin/env python3
"""Debug aid (GPU): chains of a golden set from the tile chaining path against the segment-wise kernels (AL_DBG bit 28) and the
reference's CN taps; prints the fragments whose chain multisets differ.  usage: tools/chain_diff.py <golden set> [max fragments to print]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import airlift_amd as A
from gpu_util import load_fragments, seed_lines, split_expected_seeds
import gzip, tempfile, shutil, subprocess

def unpack(name):
    src = os.path.join(ROOT, "tests", "golden", name); d = tempfile.mkdtemp(prefix="cd_")
    for f in os.listdir(src):
        if f.endswith(".gz"):
            with gzip.open(os.path.join(src, f), "rb") as i, open(os.path.join(d, f[:-3]), "wb") as o: shutil.copyfileobj(i, o)
        else: shutil.copy(os.path.join(src, f), d)
    return d

def chains(ctx, idx, nf):
    st = ctx.stat(); tot = int(st.n_anchor)
    off = ctx.tap("a_off", np.uint64, nf + 1); nu = ctx.tap("frag_nu", np.uint32, nf); na = ctx.tap("frag_na", np.uint32, nf)
    chained = ctx.tap("chained", np.uint64, tot * 2).reshape(-1, 2); u = ctx.tap("u", np.uint64, tot + nf + 1); uo = ctx.tap("uo", np.uint32, tot + nf + 1)
    out = []
    for f in range(nf):
        l = []
        for c in range(int(nu[f])):
            uc = int(u[int(off[f]) + f + c]); n = uc & 0xffffffff; k = int(off[f]) + int(uo[int(off[f]) + f + c])
            l.append((uc >> 32, int(uo[int(off[f]) + f + c]), tuple(x.split("\t", 2)[2] for x in seed_lines("CN", idx.names, chained[k:k + n], 0))))
        out.append((int(na[f]), l))
    return out

name = sys.argv[1]; maxp = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = unpack(name)
m, n_segs, seqs, names, quals = load_fragments(d)
nf = len(n_segs)
idx = A.Index(fasta=os.path.join(d, m["ref"]))
res = {}
for tag, dbg in (("tile", None), ("legacy", str(1 << 28))):
    if dbg: os.environ["AL_DBG"] = dbg
    else: os.environ.pop("AL_DBG", None)
    ctx = A.Context(idx); ctx.upload(n_segs, seqs, names); ctx.run(); res[tag] = chains(ctx, idx, nf); ctx.close()
blocks = split_expected_seeds(open(os.path.join(d, "expected.seeds")).read())
nbad = 0
for f in range(nf):
    exp, cur, last = [], [], None
    for l in blocks[f]:
        if l.startswith("CN\t"):
            cid = l.split("\t")[1]
            if cid != last and cur: exp.append(tuple(cur)); cur = []
            last = cid; cur.append(l.split("\t", 2)[2])
    if cur: exp.append(tuple(cur))
    t = sorted(c[2] for c in res["tile"][f][1]); g = sorted(c[2] for c in res["legacy"][f][1])
    if t != sorted(exp) or g != sorted(exp):
        nbad += 1
        if nbad <= maxp:
            print("== fragment %d (%d anchors): tile %s, legacy %s; chains exp %d tile %d legacy %d" % (f, res["tile"][f][0], "ok" if t == sorted(exp) else "BAD", "ok" if g == sorted(exp) else "BAD", len(exp), len(t), len(g)))
            for c in res["tile"][f][1]: print("  tile  score %d uo %d %s %s" % (c[0], c[1], "" if c[2] in exp else "NOT-IN-EXP", " | ".join(c[2])))
            for e in exp:
                if e not in t: print("  missing from tile: " + " | ".join(e))
print("%d of %d fragments differ" % (nbad, nf))
