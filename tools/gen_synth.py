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


def write_fasta(path, ref, width=60):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for name, c in ref:
            f.write(b">" + name.encode() + b"\n")
            s = lut[c]
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
    "c3": (dict(seed=2027, n_contigs=6, total_len=100_300_000, n_dups=3000, dup_len=(100, 5000), dup_div=0.03), dict(n_pairs=5_000_000, read_len=150)),
    # human-sized scale test (uniform sequence + planted diverged duplications + N blocks; not a repeat-structure model of GRCh38)
    "c4": (dict(seed=2028, n_contigs=24, total_len=3_100_000_000, n_dups=20000, dup_len=(300, 6000), dup_div=0.05, n_frac=0.02), dict(n_pairs=50_000_000, read_len=150)),
    "tiny": (dict(seed=7, n_contigs=3, total_len=300_000, n_dups=12, tandem=6), dict(n_pairs=2000, read_len=150, ins_frac=0.03)),
}


def generate(config, out, pairs=None, seed=20261002):
    os.makedirs(out, exist_ok=True)
    rk, qk = CONFIGS[config]
    qk = dict(qk)
    if pairs is not None:
        qk["n_pairs"] = pairs
    ref = make_reference(**rk)
    write_fasta(os.path.join(out, "ref.fa"), ref)
    r1, r2 = simulate_pairs(ref, seed=seed, **qk)
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
