#!/usr/bin/env python3
"""Generate the committed golden vectors from the REFERENCE build (oracle/_ref/mm2ref, mm2count).

Runs only in the authoring container (needs /root/reference to build oracle/_ref).  Inputs are made by
tools/gen_synth.py (seeded) or are the reference's own test data files; expected outputs are what the
reference build printed.  Everything written here is data (FASTA/FASTQ inputs + SAM / tap outputs), gzip'ed.

    python tests/golden/make_goldens.py
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synth as g  # noqa: E402

MM2REF = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
MM2COUNT = os.path.join(ROOT, "oracle", "_ref", "mm2count")
REFTEST = "/root/reference/src/minimap2-master_remapping/test"


def gz_write(path, data):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(data)


def run_ref(workdir, ref, reads, taps=True, rg=None):
    cmd = [MM2REF]
    if rg:
        cmd += ["-R", rg]
    sam = subprocess.run(cmd + [ref] + reads, cwd=workdir, capture_output=True, check=True).stdout
    out = {"sam": sam}
    if taps:
        sd = subprocess.run([MM2REF, "--seeds", ref] + reads, cwd=workdir, capture_output=True, check=True).stderr
        al = subprocess.run([MM2REF, "--alnseq", ref] + reads, cwd=workdir, capture_output=True, check=True).stderr
        clean = lambda b: b"".join(l + b"\n" for l in b.split(b"\n") if l and l != b"HI")
        out["seeds"] = clean(sd)
        out["alnseq"] = clean(al)
    cnt = subprocess.run([MM2COUNT, "-x", "sr", "-a", ref, reads[-1]], cwd=workdir, capture_output=True).stderr
    for l in cnt.split(b"\n"):
        if l.startswith(b"Total No. of Mappings"):
            out["count"] = int(l.split(b":")[-1])
    return out


def emit(name, workdir, ref, reads, taps=True, rg=None, keep_inputs=True, meta=None):
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    res = run_ref(workdir, ref, reads, taps, rg)
    if keep_inputs:
        for fn in [ref] + reads:
            gz_write(os.path.join(d, os.path.basename(fn) + ".gz"), open(os.path.join(workdir, fn), "rb").read())
    gz_write(os.path.join(d, "expected.sam.gz"), res["sam"])
    if taps:
        gz_write(os.path.join(d, "expected.seeds.gz"), res["seeds"])
        gz_write(os.path.join(d, "expected.alnseq.gz"), res["alnseq"])
    m = {"ref": os.path.basename(ref), "reads": [os.path.basename(r) for r in reads], "count": res.get("count"),
         "rg": rg, "n_sam_lines": res["sam"].count(b"\n"), "sam_md5": hashlib.md5(res["sam"]).hexdigest()}
    if meta:
        m.update(meta)
    json.dump(m, open(os.path.join(d, "meta.json"), "w"), indent=1, sort_keys=True)
    print(name, m["n_sam_lines"], "SAM lines, count =", m["count"])


def main():
    tmp = tempfile.mkdtemp(prefix="al_gold_")
    # G1: reference's own MT-human.fa as the reference + 1000 simulated 150 bp pairs
    shutil.copy(os.path.join(REFTEST, "MT-human.fa"), os.path.join(tmp, "MT-human.fa"))
    seq = "".join(l.strip() for l in open(os.path.join(tmp, "MT-human.fa")) if not l.startswith(">"))
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in seq], dtype=np.uint8)
    mt = [("MT_human", codes)]
    r1, r2 = g.simulate_pairs(mt, 1000, 150, seed=101, ins_mean=400, ins_sd=40)
    g.write_fastq(os.path.join(tmp, "g1_1.fq"), r1); g.write_fastq(os.path.join(tmp, "g1_2.fq"), r2)
    emit("g1_mt150pe", tmp, "MT-human.fa", ["g1_1.fq", "g1_2.fq"], rg="@RG\\tID:S1\\tSM:S1\\tPL:illumina\\tLB:S1")
    # G2: 100 bp SE (config 1 shape) and 250 bp PE (config 5 shape)
    ref = g.make_reference(seed=11, n_contigs=4, total_len=400_000, n_dups=40, dup_len=(300, 3000), dup_div=0.01, tandem=30, n_frac=0.002)
    g.write_fasta(os.path.join(tmp, "syn.fa"), ref)
    se, _ = g.simulate_pairs(ref, 1500, 100, seed=7, single_end=True, sub_rate=0.01, del_frac=0.1)
    g.write_fastq(os.path.join(tmp, "g2_se.fq"), se, prefix="realigned_singleton_")
    emit("g2_100se", tmp, "syn.fa", ["g2_se.fq"])
    r1, r2 = g.simulate_pairs(ref, 800, 250, seed=6, ins_mean=550, ins_sd=60, ins_lo=260, ins_hi=780, sub_rate=0.01, del_frac=0.2, ins_frac=0.2)
    g.write_fastq(os.path.join(tmp, "g2_1.fq"), r1); g.write_fastq(os.path.join(tmp, "g2_2.fq"), r2)
    emit("g2_250pe", tmp, "syn.fa", ["g2_1.fq", "g2_2.fq"], keep_inputs=True)
    # G3: adversarial 150 bp PE: overlapping mates (heap ties), 2 % subs, indels, N, tandem repeats, random reads,
    #     reads at contig ends, indels near read ends, very short reads
    r1, r2 = g.simulate_pairs(ref, 1500, 150, seed=5, ins_mean=250, ins_sd=60, ins_lo=150, ins_hi=700, sub_rate=0.02, del_frac=0.3, ins_frac=0.3, n_rate=0.003)
    rng = np.random.default_rng(3)
    r1[-60:] = rng.integers(0, 4, size=(60, 150)); r2[-30:] = rng.integers(0, 4, size=(30, 150))
    seqs1 = [bytes(b"ACGTN"[c] for c in row) for row in r1]
    seqs2 = [bytes(b"ACGTN"[c] for c in row) for row in r2]
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    big = ref[0][1]
    extra1, extra2 = [], []
    for i in range(40):  # contig ends + indels at read ends + ragged lengths
        L = int(rng.integers(30, 151))
        s = 0 if i % 2 == 0 else len(big) - L - 300
        a = lut[big[s:s + L]].tobytes()
        b = lut[g.revcomp(big[s + 200:s + 200 + L])].tobytes()
        if i % 4 == 1:
            a = a[:4] + a[6:]          # deletion 4 bp from the read start
        if i % 4 == 2:
            b = b[:-5] + b"GT" + b[-5:]  # insertion near the read end
        if i % 8 == 3:
            a = a[:10] + b"N" * 5 + a[15:]
        extra1.append(a); extra2.append(b)
    extra1 += [b"ACGT" * 5, b"A" * 150, b"AC" * 75, b"N" * 40, b"ACGTTGCA" * 18 + b"ACGTTG"]
    extra2 += [b"TTGACCA", b"T" * 150, b"GT" * 75, b"ACGTN" * 30, b"TGCAACGT" * 18 + b"TGCAAC"]
    s1 = seqs1 + extra1; s2 = seqs2 + extra2
    names = ["realigned_%d" % i for i in range(len(s1))]
    g.write_fastq_ragged(os.path.join(tmp, "g3_1.fq"), s1, [n + "/1" for n in names])
    g.write_fastq_ragged(os.path.join(tmp, "g3_2.fq"), s2, [n + "/2" for n in names])
    emit("g3_adversarial", tmp, "syn.fa", ["g3_1.fq", "g3_2.fq"], rg="@RG\\tID:adv\\tSM:adv")
    # G4: the reference's own fixtures (long-ish reads through the same sr path; q2 is unmapped => FLAG 4 golden)
    for t, q in (("MT-human.fa", "MT-orang.fa"), ("t-inv.fa", "q-inv.fa"), ("t2.fa", "q2.fa")):
        shutil.copy(os.path.join(REFTEST, t), os.path.join(tmp, t)); shutil.copy(os.path.join(REFTEST, q), os.path.join(tmp, q))
        emit("g4_" + q.split(".")[0].replace("-", "_"), tmp, t, [q], taps=(q != "MT-orang.fa"))
    # G6: repeat-rich reference (a 300 bp element in ~2600 copies, 0-3 % divergence): minimizers above mid_occ = 1000, so
    #     rep_len > 0, the max_occ re-chain pass (map.c:353-375), thousands of anchors per fragment.  SAM + count only.
    rng = np.random.default_rng(77)
    rep = g.make_reference(seed=21, n_contigs=2, total_len=1_200_000, n_dups=20, dup_len=(300, 2000), dup_div=0.01)
    unit = rng.integers(0, 4, size=300, dtype=np.uint8)
    for _, c in rep:
        for _ in range(1300):
            s = int(rng.integers(0, len(c) - 300))
            u = unit.copy(); m = rng.random(300) < rng.uniform(0, 0.03)
            u[m] = (u[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            c[s:s + 300] = u
    g.write_fasta(os.path.join(tmp, "rep.fa"), rep)
    r1, r2 = g.simulate_pairs(rep, 500, 150, seed=9, ins_mean=380, ins_sd=50, sub_rate=0.005, del_frac=0.05)
    g.write_fastq(os.path.join(tmp, "g6_1.fq"), r1); g.write_fastq(os.path.join(tmp, "g6_2.fq"), r2)
    emit("g6_repeats", tmp, "rep.fa", ["g6_1.fq", "g6_2.fq"], taps=False)
    # G5: yeast-sized, only digests committed (inputs are regenerated by tools/gen_synth.py at test time)
    d5 = os.path.join(tmp, "g5")
    g.generate("c2", d5, pairs=100_000)
    sam = subprocess.run([MM2REF, "-t", "8", "ref.fa", "reads_1.fq", "reads_2.fq"], cwd=d5, capture_output=True, check=True).stdout
    cols = b"\n".join(b"\t".join(l.split(b"\t")[:9]) for l in sam.split(b"\n") if l and not l.startswith(b"@"))
    os.makedirs(os.path.join(HERE, "g5_yeast100k"), exist_ok=True)
    json.dump({"config": "c2", "pairs": 100_000, "seed": 20261002, "cols1_9_md5": hashlib.md5(cols).hexdigest(),
               "sam_md5": hashlib.md5(sam).hexdigest(), "n_records": cols.count(b"\n") + 1},
              open(os.path.join(HERE, "g5_yeast100k", "meta.json"), "w"), indent=1, sort_keys=True)
    print("g5", hashlib.md5(cols).hexdigest())
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
