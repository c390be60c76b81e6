#!/usr/bin/env python3
"""Golden vectors for the token stage: the reference's own src/2-generate_gaps/gaps_to_fasta.py tiles a gap
FASTA into read-sized tokens, the reference build (oracle/_ref/mm2ref) aligns the tokens single-end.  Authoring container only
(needs /root/reference); what is committed is data: the gap FASTA, the target FASTA (the fork's own test/MT-human.fa) and the SAM.

    python tests/golden/make_g7_tokens.py
"""
import gzip, hashlib, json, os, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
MM2REF = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
SCRIPT = "/root/reference/src/2-generate_gaps/gaps_to_fasta.py"
MT = "/root/reference/src/minimap2-master_remapping/test/MT-human.fa"
READ_SIZE, SKIP = 100, 7


def gz_write(path, data):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(data)


def main():
    tmp = tempfile.mkdtemp(prefix="al_g7_")
    seq = "".join(l.strip() for l in open(MT) if not l.startswith(">"))
    rng = np.random.default_rng(123)
    recs = []
    for i, (s, L) in enumerate([(100, 700), (3000, 250), (5200, 100), (6000, 101), (7000, 99), (9000, 1500), (12000, 107), (15000, 333)]):
        t = list(seq[s:s + L])
        for p in rng.integers(0, L, size=max(1, L // 60)):          # a few substitutions
            t[int(p)] = "ACGT"[int(rng.integers(0, 4))]
        if i == 1:
            t[40:44] = list("nnNN")                                     # N run, lower case
        if i == 5:
            t = [c.lower() if j % 3 == 0 else c for j, c in enumerate(t)]; t[700] = "U"; del t[900:905]   # mixed case, a U, a deletion
        recs.append(("MT_human:%d-%d" % (s, s + L), "".join(t)))
    with open(os.path.join(tmp, "gaps.fa"), "w") as f:
        for i, (n, t) in enumerate(recs):
            f.write(">" + n + "\n")
            w = 60 if i % 2 == 0 else 10_000                         # multi-line and single-line records
            for o in range(0, len(t), w):
                f.write(t[o:o + w] + "\n")
    # (the script ends with an IndexError on its last statement after all tokens are written: exit status ignored)
    subprocess.run([sys.executable, SCRIPT, "gaps.fa", str(READ_SIZE), "tokens.fa", str(SKIP)], cwd=tmp, check=False, stderr=subprocess.DEVNULL)
    sam = subprocess.run([MM2REF, MT, "tokens.fa"], cwd=tmp, capture_output=True, check=True).stdout
    d = os.path.join(HERE, "g7_tokens"); os.makedirs(d, exist_ok=True)
    gz_write(os.path.join(d, "gaps.fa.gz"), open(os.path.join(tmp, "gaps.fa"), "rb").read())
    gz_write(os.path.join(d, "MT-human.fa.gz"), open(MT, "rb").read())
    gz_write(os.path.join(d, "expected.sam.gz"), sam)
    n_tok = sum(1 for l in open(os.path.join(tmp, "tokens.fa")) if l.startswith(">"))
    json.dump({"ref": "MT-human.fa", "gaps": "gaps.fa", "read_size": READ_SIZE, "skip": SKIP, "n_tokens": n_tok, "n_sam_lines": sam.count(b"\n"),
               "sam_md5": hashlib.md5(sam).hexdigest()}, open(os.path.join(d, "meta.json"), "w"), indent=1, sort_keys=True)
    print("g7_tokens:", n_tok, "tokens,", sam.count(b"\n"), "SAM lines")


if __name__ == "__main__":
    main()
