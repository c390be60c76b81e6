#!/usr/bin/env python3
"""Deterministic synthetic reference + read generator.

Pure numpy; no reference code involved.  Used by tests (small cases), by the golden-vector
script (tests/golden/make_goldens.py) and by bench.py (C2 workload).

    python tools/gen_synth.py --config c2 --out /tmp/c2 [--pairs N]
"""
import argparse
import os
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def make_reference(seed, n_contigs, total_len, n_dups=0, dup_len=(2000, 6000), dup_div=0.02,
                   n_frac=0.0, tandem=0, family=None):
    """Uniform ACGT contigs with planted diverged duplications (and optional N blocks / tandem repeats)."""
    rng = np.random.default_rng(seed)
    clen = total_len // n_contigs
    contigs = [rng.integers(0, 4, size=clen, dtype=np.uint8) for _ in range(n_contigs)]
    for _ in range(n_dups):
        L = int(rng.integers(dup_len[0], dup_len[1] + 1))
        a, b = rng.integers(0, n_contigs, size=2)
        if clen <= L + 2:
            continue
        s = int(rng.integers(0, clen - L)); d = int(rng.integers(0, clen - L))
        seg = contigs[a][s:s + L].copy()
        m = rng.random(L) < dup_div
        seg[m] = (seg[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        contigs[b][d:d + L] = seg
    for _ in range(tandem):
        c = int(rng.integers(0, n_contigs)); unit = int(rng.integers(2, 40)); cn = int(rng.integers(5, 30))
        if clen <= unit * cn + 2:
            continue
        s = int(rng.integers(0, clen - unit * cn))
        u = rng.integers(0, 4, size=unit, dtype=np.uint8)
        contigs[c][s:s + unit * cn] = np.tile(u, cn)
    if family:   # an interspersed high-copy element family (copies, unit length, divergence per copy), Alu-like
        copies, ulen, div = family
        unit = rng.integers(0, 4, size=ulen, dtype=np.uint8)
        for _ in range(copies):
            c = contigs[int(rng.integers(0, n_contigs))]
            if len(c) <= ulen + 2:
                continue
            p0 = int(rng.integers(0, len(c) - ulen)); u = unit.copy(); m = rng.random(ulen) < div
            u[m] = (u[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            c[p0:p0 + ulen] = u
    if n_frac > 0:
        for c in contigs:
            L = int(len(c) * n_frac)
            if L > 0:
                s = int(rng.integers(0, len(c) - L)); c[s:s + L] = 4
    return [("chr%d" % (i + 1), c) for i, c in enumerate(contigs)]


# ---------------------------------------------------------------------------------------------------------------
# SURVEY.md 8(d) reference models C3 / C4 / C5: interspersed repeat families + N blocks on uniform sequence.
# Deterministic for a given seed whatever the thread count (one RNG stream per contig, families from a master stream).
# ---------------------------------------------------------------------------------------------------------------
def _rand_bases(bitgen, n):
    """n uniform bases (uint8 codes 0..3) from raw 64-bit draws (two of every eight bits used)."""
    return (bitgen.random_raw((n + 7) // 8).view(np.uint8) & 3)[:n]


def _rand_u8(bitgen, shape):
    n = int(np.prod(shape))
    return bitgen.random_raw((n + 7) // 8).view(np.uint8)[:n].reshape(shape)


def family_plan(seed, total_len, model):
    """List of (unit_len, copies, divergence) for a repeat model.
    'human': 200 families to 45 % of the sequence -- 40 SINE-like (about 300 bp, up to 1e5 copies), 100 LINE-like (about 6 kb,
    up to 1e3 copies, most copies 5'-truncated), 60 of intermediate size; divergence of a family uniform in 0-15 %.
    'worm': 100-5000 bp units, 2-50 copies, 1-5 % divergence, to 3 % of the sequence.
    Copy numbers are scaled with total_len (the class maxima are those of a 3.1 Gbp genome), so small test references keep
    the same repeat density and the same per-minimizer occurrence profile relative to their size."""
    rng = np.random.default_rng([seed, 77])
    fams = []
    if model == "human":
        scale = total_len / 3.1e9
        for j in range(200):
            if j < 40: L = int(rng.integers(280, 321)); c = np.exp(rng.uniform(np.log(2e4), np.log(1e5))); cmax = 1e5
            elif j < 140: L = int(rng.integers(5000, 6501)); c = np.exp(rng.uniform(np.log(200), np.log(1e3))); cmax = 1e3
            else: L = int(rng.integers(800, 3001)); c = np.exp(rng.uniform(np.log(500), np.log(2e4))); cmax = 2e4
            fams.append([L, c, float(rng.uniform(0.0, 0.15)), cmax])
        target = 0.598 * 3.1e9      # inserted bases; copies overlap at random, so 1 - exp(-0.598) = 45 % of the sequence is covered
        for _ in range(20):   # scale copy numbers to 45 % (average copy length of a truncated family is ~0.5 L), class maxima kept
            tot = sum(L * c * (0.5 if L >= 800 else 1.0) for L, c, _, _ in fams)
            for f in fams: f[1] = min(f[3], f[1] * target / tot)
        return [(L, max(2, int(round(c * scale))), d) for L, c, d, _ in fams]
    if model == "worm":
        tot = 0
        while tot < 0.03 * total_len:
            L = int(rng.integers(100, 5001)); c = int(rng.integers(2, 51)); d = float(rng.uniform(0.01, 0.05))
            fams.append((L, c, d)); tot += L * c
        return fams
    raise ValueError(model)


def make_reference_model(seed, n_contigs, total_len, model, n_frac=0.0, threads=None, count_covered=False):
    """Uniform ACGT contigs + the interspersed repeat families of family_plan() + N blocks (one centromere-like block of
    0.6 n_frac per contig and ten smaller gaps).  Returns [(name, uint8 codes)] whose arrays are views of one buffer."""
    from concurrent.futures import ThreadPoolExecutor
    fams = family_plan(seed, total_len, model)
    master = np.random.default_rng([seed, 78])
    # human-like contig lengths: a geometric-ish spread (chr1 is ~5x chr21), C3: nearly equal
    wts = np.linspace(1.0, 0.2, n_contigs) if model == "human" else np.ones(n_contigs)
    lens = np.floor(wts / wts.sum() * total_len).astype(np.int64); lens[0] += total_len - lens.sum()
    cons = [_rand_bases(np.random.PCG64([seed, 79, j]), L) for j, (L, _, _) in enumerate(fams)]
    # copies of every family spread over the contigs in proportion to their length
    per = [master.multinomial(c, lens / lens.sum()) for _, c, _ in fams]
    big = np.empty(int(lens.sum()), dtype=np.uint8)
    cum = np.concatenate([[0], np.cumsum(lens)])

    inserted = np.zeros(n_contigs, dtype=np.int64); covered = np.zeros(n_contigs, dtype=np.int64)

    def build(ci):
        mask = np.zeros(int(lens[ci]), dtype=bool) if count_covered else None
        bg = np.random.PCG64([seed, 80, ci]); rng = np.random.Generator(np.random.PCG64([seed, 81, ci]))
        n = int(lens[ci]); c = big[cum[ci]:cum[ci + 1]]
        c[:] = _rand_bases(bg, n)
        for j, (L, _, d) in enumerate(fams):
            k = int(per[j][ci])
            if k == 0 or n <= L + 2: continue
            thr = int(round(d * 256))
            for s in range(0, k, 4096):      # copies in slabs (memory)
                m = min(4096, k - s)
                cp = np.broadcast_to(cons[j], (m, L)).copy()
                if thr > 0:
                    mut = _rand_u8(bg, (m, L)) < thr
                    nm = int(mut.sum())
                    cp[mut] = (cp[mut] + (_rand_u8(bg, nm) % 3 + 1).astype(np.uint8)) & 3
                p0 = rng.integers(0, n - L, size=m)
                rev = rng.random(m) < 0.5
                col = np.arange(L)[None, :]
                if L >= 800 and model == "human":   # LINE-like: most copies keep only a 3' part
                    keep = np.where(rng.random(m) < 0.75, np.maximum(100, (L * rng.random(m) ** 2).astype(np.int64)), L)
                else: keep = np.full(m, L, dtype=np.int64)
                off = col - (L - keep)[:, None]                       # position inside the kept 3' part
                valid = off >= 0
                off = np.where(rev[:, None], keep[:, None] - 1 - off, off)
                val = np.where(rev[:, None], 3 - cp, cp)
                dst = (p0[:, None] + off)[valid]
                c[dst] = val[valid]                                    # later copies overwrite earlier ones where they overlap
                inserted[ci] += len(dst)
                if mask is not None: mask[dst] = True
        if n_frac > 0:
            Lc = int(n * n_frac * 0.6)
            if Lc > 0: s0 = int(rng.integers(n // 4, n // 2)); c[s0:s0 + Lc] = 4
            Lg = int(n * n_frac * 0.04)
            for _ in range(10):
                if Lg > 0: s0 = int(rng.integers(0, n - Lg)); c[s0:s0 + Lg] = 4
        if mask is not None: covered[ci] = int((mask & (c < 4)).sum())
        return ci

    with ThreadPoolExecutor(max_workers=threads or min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(build, range(n_contigs)))
    ref = RefList(("chr%d" % (i + 1), big[cum[i]:cum[i + 1]]) for i in range(n_contigs))
    ref.big = big
    ref.stats = {"families": len(fams), "inserted_frac": float(inserted.sum()) / float(lens.sum()),
                 "covered_frac": float(covered.sum()) / float(lens.sum()) if count_covered else None, "max_copies": max(c for _, c, _ in fams)}
    return ref


class RefList(list):
    """[(name, codes)] plus .big = the concatenation of all contigs (the arrays are views of it)."""
    big = None
    stats = None


def write_fasta(path, ref, width=60):
    tr = bytes.maketrans(bytes(range(5)), b"ACGTN")
    with open(path, "wb") as f:
        for name, c in ref:
            f.write(b">" + name.encode() + b"\n")
            s = np.frombuffer(c.tobytes().translate(tr), dtype=np.uint8)
            n = len(s); full = n // width * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = s[:full].reshape(-1, width); body[:, width] = 10
                f.write(body.tobytes())
            if n > full:
                f.write(s[full:].tobytes() + b"\n")


def revcomp(a):
    return COMP[a[..., ::-1]]


def simulate_pairs(ref, n_pairs, read_len, seed, ins_mean=400, ins_sd=40, ins_lo=160, ins_hi=780,
                   sub_rate=0.005, del_frac=0.05, del_len=2, ins_frac=0.0, n_rate=0.0, single_end=False):
    """Returns (r1, r2) uint8 code arrays [n_pairs, read_len] in sequencing orientation (FR pairs)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for _, c in ref]); cum = np.concatenate([[0], np.cumsum(lens)])
    ins = np.clip(np.rint(rng.normal(ins_mean, ins_sd, n_pairs)).astype(np.int64), max(ins_lo, read_len), ins_hi)
    if single_end:
        ins[:] = read_len
    ci = rng.integers(0, len(ref), n_pairs)
    st = (rng.random(n_pairs) * (lens[ci] - ins - del_len - 1)).astype(np.int64)
    big = np.concatenate([c for _, c in ref])
    g0 = cum[ci] + st
    idx = np.arange(read_len)[None, :]
    # mate 1: forward strand at fragment start; optional small deletion in the middle
    has_del = rng.random(n_pairs) < del_frac
    dpos = rng.integers(read_len // 4, 3 * read_len // 4, n_pairs)
    shift = np.where(has_del[:, None] & (idx >= dpos[:, None]), del_len, 0)
    r1 = big[g0[:, None] + idx + shift]
    # mate 2: reverse complement of the fragment end
    e0 = g0 + ins - read_len
    r2 = revcomp(big[e0[:, None] + idx])
    if ins_frac > 0:  # small insertion in mate 2
        has_ins = rng.random(n_pairs) < ins_frac
        ipos = rng.integers(read_len // 4, 3 * read_len // 4, n_pairs)
        for i in np.nonzero(has_ins)[0]:
            p = int(ipos[i]); row = r2[i].copy()
            r2[i, p + 2:] = row[p:read_len - 2]; r2[i, p:p + 2] = rng.integers(0, 4, 2)
    for r in (r1, r2):
        m = (rng.random(r.shape) < sub_rate) & (r < 4)
        r[m] = (r[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        if n_rate > 0:
            r[rng.random(r.shape) < n_rate] = 4
    # random fragment strand: swap roles so that mate 1 is sometimes the reverse-strand read
    flip = rng.random(n_pairs) < 0.5
    r1f = np.where(flip[:, None], r2, r1); r2f = np.where(flip[:, None], r1, r2)
    return r1f, r2f


def simulate_pairs_mason(ref, n_pairs, read_len, seed, ins_mean=400, ins_sd=40, snp=1e-3, indel=2e-4, indel_max=10,
                         err_lo=0.002, err_hi=0.01):
    """Paired-end reads in the manner of the mason2 recipe the fork's paper uses (tex/hs38-simu.sh:9-10): the sampled haplotype
    differs from the reference by SNPs (1e-3 per base) and small indels (2e-4 per base, 1..indel_max bp; at most one per mate
    here), sequencing errors are substitutions whose rate rises linearly from err_lo at the first cycle to err_hi at the last.
    Fragments are not drawn from positions that start or end in an N block.  Returns (r1, r2) code arrays in sequencing orientation."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for _, c in ref]); cum = np.concatenate([[0], np.cumsum(lens)])
    big = getattr(ref, "big", None)
    if big is None: big = np.concatenate([c for _, c in ref])
    ins = np.clip(np.rint(rng.normal(ins_mean, ins_sd, n_pairs)).astype(np.int64), read_len + indel_max, None)
    pad = indel_max + 1
    ci = rng.choice(len(ref), size=n_pairs, p=lens / lens.sum())
    g0 = cum[ci] + (rng.random(n_pairs) * (lens[ci] - ins - 2 * pad)).astype(np.int64) + pad
    for _ in range(12):   # re-draw fragments that touch an N block at either end or in the middle
        bad = (big[g0] == 4) | (big[g0 + ins - 1] == 4) | (big[g0 + ins // 2] == 4) | (big[g0 + read_len - 1] == 4) | (big[g0 + ins - read_len] == 4)
        nb = int(bad.sum())
        if nb == 0: break
        c2 = rng.choice(len(ref), size=nb, p=lens / lens.sum())
        ci[bad] = c2; g0[bad] = cum[c2] + (rng.random(nb) * (lens[c2] - ins[bad] - 2 * pad)).astype(np.int64) + pad
    idx = np.arange(read_len)[None, :]

    win = np.lib.stride_tricks.sliding_window_view(big, read_len + indel_max)

    def mate(start):      # read_len bases of the haplotype starting at reference offset `start`
        has = rng.random(n_pairs) < indel * read_len
        is_del = rng.random(n_pairs) < 0.5
        ln = rng.integers(1, indel_max + 1, n_pairs)
        p = rng.integers(read_len // 8, 7 * read_len // 8, n_pairs)
        w = win[start]                                            # one contiguous row copy per read
        r = np.ascontiguousarray(w[:, :read_len])
        h = np.nonzero(has)[0]
        if len(h):
            d = ln[h][:, None]; ph = p[h][:, None]; dl = is_del[h][:, None]
            # deletion of d reference bases at p: later bases come from d further on; insertion of d random bases at p: later bases from d back
            shift = np.where(dl, np.where(idx >= ph, d, 0), np.where(idx >= ph + d, -d, 0))
            rh = np.take_along_axis(w[h], idx + shift, axis=1)
            insm = (~dl) & (idx >= ph) & (idx < ph + d)
            k = int(insm.sum())
            if k: rh[insm] = rng.integers(0, 4, size=k, dtype=np.uint8)
            r[h] = rh
        return r

    r1 = mate(g0)
    r2 = mate(g0 + ins - read_len)
    def substitute(r, colp):   # substitutions at per-column rates colp: number of events drawn first, then their places
        k = int(rng.binomial(r.size, float(colp.mean())))
        rows = rng.integers(0, r.shape[0], k); cols = rng.choice(read_len, size=k, p=colp / colp.sum())
        v = r[rows, cols]; ok = v < 4
        r[rows[ok], cols[ok]] = (v[ok] + rng.integers(1, 4, size=int(ok.sum()), dtype=np.uint8)) & 3

    for r in (r1, r2): substitute(r, np.full(read_len, snp))                     # haplotype SNPs
    r2 = revcomp(r2)
    ramp = err_lo + (err_hi - err_lo) * np.arange(read_len) / max(1, read_len - 1)
    for r in (r1, r2): substitute(r, ramp)                                        # sequencing errors, by cycle
    flip = rng.random(n_pairs) < 0.5
    return np.where(flip[:, None], r2, r1), np.where(flip[:, None], r1, r2)


def write_fastq(path, reads, prefix="realigned_", suffix="", qual=b"I", names=None):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    n, L = reads.shape
    with open(path, "wb") as f:
        q = qual * L
        chunk = []
        for i in range(n):
            nm = names[i] if names is not None else ("%s%d%s" % (prefix, i, suffix))
            chunk.append(b"@" + nm.encode() + b"\n" + lut[reads[i]].tobytes() + b"\n+\n" + q + b"\n")
            if len(chunk) >= 65536:
                f.write(b"".join(chunk)); chunk = []
        f.write(b"".join(chunk))


def write_fastq_ragged(path, seqs, names, qual=b"I"):
    with open(path, "wb") as f:
        for nm, s in zip(names, seqs):
            f.write(b"@" + nm.encode() + b"\n" + s + b"\n+\n" + qual * len(s) + b"\n")


CONFIGS = {
    # name: (ref kwargs, read kwargs)
    "c1": (dict(seed=2026, n_contigs=16, total_len=12_160_000, n_dups=150), dict(n_pairs=100_000, read_len=100, single_end=True, del_frac=0.0)),
    "c2": (dict(seed=2026, n_contigs=16, total_len=12_160_000, n_dups=150), dict(n_pairs=2_000_000, read_len=150)),
    # C2 plus an interspersed element family covering ~10 % of the reference (4000 copies x 300 bp, 10 % divergence per copy):
    # minimizers above mid_occ, the max_occ re-chain pass, thousands of anchors for the fragments inside copies
    "c2r": (dict(seed=2026, n_contigs=16, total_len=12_160_000, n_dups=150, family=(4000, 300, 0.10)), dict(n_pairs=2_000_000, read_len=150)),
    # ce11-sized, 3 % of the sequence in 100-5000 bp repeats (copy number 2-50, 1-5 % divergence)
    "c3": (dict(model="worm", seed=2027, n_contigs=6, total_len=100_300_000), dict(n_pairs=5_000_000, read_len=150, mason=True)),
    # GRCh38-sized, 24 contigs, 3.1 Gbp, 5 % N blocks, 45 % repeat content from 200 families (Alu-like 300 bp up to
    # 1e5 copies, L1-like 6 kb up to 1e3 copies, 0-15 % divergence); mason-like reads (tex/hs38-simu.sh:9-10)
    "c4": (dict(model="human", seed=2028, n_contigs=24, total_len=3_100_000_000, n_frac=0.05), dict(n_pairs=50_000_000, read_len=150, mason=True)),
    # C5: the C4 reference, 250 bp PE, insert N(550, 60)
    "c5": (dict(model="human", seed=2028, n_contigs=24, total_len=3_100_000_000, n_frac=0.05), dict(n_pairs=50_000_000, read_len=250, mason=True, ins_mean=550, ins_sd=60)),
    # the C4 model at 1/10 of the size (same repeat density; copy numbers scale with the length): fits small hosts
    "c4s": (dict(model="human", seed=2028, n_contigs=24, total_len=310_000_000, n_frac=0.05), dict(n_pairs=5_000_000, read_len=150, mason=True)),
    # round-1 scale tests without a repeat model (uniform sequence + planted duplications)
    "c3u": (dict(seed=2027, n_contigs=6, total_len=100_300_000, n_dups=3000, dup_len=(100, 5000), dup_div=0.03), dict(n_pairs=5_000_000, read_len=150)),
    "c4u": (dict(seed=2028, n_contigs=24, total_len=3_100_000_000, n_dups=20000, dup_len=(300, 6000), dup_div=0.05, n_frac=0.02), dict(n_pairs=50_000_000, read_len=150)),
    "tiny": (dict(seed=7, n_contigs=3, total_len=300_000, n_dups=12, tandem=6), dict(n_pairs=2000, read_len=150, ins_frac=0.03)),
}


def build_reference(config, cache_dir=None, **kw):
    """Reference of a config.  cache_dir (or $AL_REF_CACHE): keep / reuse the generated contigs as one .npy file, so that
    several runs on one machine (bench, profiles, tests) pay the generation once; the content is a pure function of the config."""
    rk = dict(CONFIGS[config][0])
    if "model" not in rk:
        return make_reference(**rk)
    cache_dir = cache_dir or os.environ.get("AL_REF_CACHE")
    tag = "%s_%d_%d_%d_%s" % (rk["model"], rk["seed"], rk["n_contigs"], rk["total_len"], rk.get("n_frac", 0.0))
    if cache_dir:
        import json
        fn = os.path.join(cache_dir, "ref_" + tag + ".npy"); fm = fn + ".json"
        if os.path.exists(fn) and os.path.exists(fm):
            meta = json.load(open(fm)); big = np.load(fn); cum = np.concatenate([[0], np.cumsum(meta["lens"])])
            ref = RefList(("chr%d" % (i + 1), big[cum[i]:cum[i + 1]]) for i in range(len(meta["lens"])))
            ref.big = big; ref.stats = meta["stats"]
            return ref
    ref = make_reference_model(**rk, **kw)
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        np.save(fn + ".tmp.npy", ref.big); os.replace(fn + ".tmp.npy", fn)
        json.dump({"lens": [int(len(c)) for _, c in ref], "stats": ref.stats}, open(fm, "w"))
    return ref


def simulate(config, ref, n_pairs, seed, read_len=None, **over):
    """Reads of a config's read model (mason-like for C3..C5, the simple model of round 1 for the others)."""
    qk = dict(CONFIGS[config][1]); qk.pop("n_pairs", None); qk.update(over)
    rl = read_len or qk.pop("read_len"); qk.pop("read_len", None)
    if qk.pop("mason", False):
        if rl >= 200 and "ins_mean" not in qk: qk.update(ins_mean=550, ins_sd=60)
        return simulate_pairs_mason(ref, n_pairs, rl, seed=seed, **qk)
    if rl >= 200 and "ins_mean" not in qk: qk.update(ins_mean=550, ins_sd=60, ins_hi=1000)
    return simulate_pairs(ref, n_pairs, rl, seed=seed, **qk)


def generate(config, out, pairs=None, seed=20261002):
    os.makedirs(out, exist_ok=True)
    rk, qk = CONFIGS[config]
    qk = dict(qk)
    if pairs is not None:
        qk["n_pairs"] = pairs
    ref = build_reference(config)
    write_fasta(os.path.join(out, "ref.fa"), ref)
    r1, r2 = simulate(config, ref, qk["n_pairs"], seed)
    if qk.get("single_end"):
        write_fastq(os.path.join(out, "reads.fq"), r1)
    else:
        write_fastq(os.path.join(out, "reads_1.fq"), r1)
        write_fastq(os.path.join(out, "reads_2.fq"), r2)
    return ref, r1, r2


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--out", required=True)
    ap.add_argument("--pairs", type=int, default=None)
    a = ap.parse_args()
    generate(a.config, a.out, a.pairs)
