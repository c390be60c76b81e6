import os, sys, subprocess, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import load_fragments, seed_lines, split_expected_seeds
import airlift_amd as A
d = '/tmp/g6'
exp = subprocess.run(['/root/repo/oracle/al_oracle', '--seeds', 'rep.fa', 'g6_1.fq', 'g6_2.fq'], cwd=d, capture_output=True).stderr.decode()
blocks = split_expected_seeds(exp)
m, n_segs, seqs, names, quals = load_fragments(d)
idx = A.Index(fasta=os.path.join(d, m['ref']))
if os.environ.get('NORECHAIN'): idx.mo.max_occ = idx.mo.mid_occ
ctx = A.Context(idx); ctx.upload(n_segs, seqs, names); ctx.run()
nf = len(n_segs); st = ctx.stat(); tot = int(st.n_anchor)
na1 = ctx.tap('frag_na_p1', np.uint32, nf); off1 = ctx.tap('a_off_p1', np.uint64, nf + 1); rep1 = ctx.tap('frag_rep_p1', np.int32, nf)
off = ctx.tap('a_off', np.uint64, nf + 1); nu = ctx.tap('frag_nu', np.uint32, nf); na = ctx.tap('frag_na', np.uint32, nf)
anchors = ctx.tap('anchors', np.uint64, tot * 2).reshape(-1, 2); chained = ctx.tap('chained', np.uint64, tot * 2).reshape(-1, 2)
u = ctx.tap('u', np.uint64, tot + nf + 1)
print('n_rechain', st.n_rechain, 'heap_fallback', st.n_heap_fallback, 'tot anchors', tot)
bad_sd = bad_cn = 0
for f in range(nf):
    e = blocks[f]
    got = ['RS\t%d' % rep1[f]] + seed_lines('SD', idx.names, anchors[int(off1[f]):int(off1[f]) + int(na1[f])])
    if got != [l for l in e if not l.startswith('CN\t')]:
        bad_sd += 1
        if bad_sd <= 3: print('SD mismatch frag', f, 'na1', na1[f], 'exp', len(e))
    exp_chains, cur, last = [], [], None
    for l in e:
        if l.startswith('CN\t'):
            cid = l.split('\t')[1]
            if cid != last and cur: exp_chains.append(tuple(cur)); cur = []
            last = cid; cur.append(l.split('\t', 2)[2])
    if cur: exp_chains.append(tuple(cur))
    uu = u[int(off[f]) + f:int(off[f]) + f + int(nu[f])]; k = int(off[f]); gc = []
    for c in range(int(nu[f])):
        n = int(uu[c] & np.uint64(0xffffffff)); gc.append(tuple(l.split('\t', 2)[2] for l in seed_lines('CN', idx.names, chained[k:k + n], 0))); k += n
    if sorted(gc) != sorted(exp_chains):
        bad_cn += 1
        if bad_cn <= 6: print('CN mismatch frag', f, 'na', na[f], 'na1', na1[f], 'nu', nu[f], 'exp chains', len(exp_chains), 'rep', rep1[f])
print('bad SD', bad_sd, 'bad CN', bad_cn, 'of', nf)
