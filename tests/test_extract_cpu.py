"""CPU tests of SURVEY.md N1 (`airlift-align extract-reads` / `extract-sequence`, al_extract.cpp) against the Python restatement
of the reference's scripts (oracle/n1_oracle.py: parity unpinned for the samtools / bedops / BBMap parts, which are not in this
image) and, for the FASTQ subset step, against seqtk 1.3 itself as vendored by the reference (oracle/_ref/seqtk)."""
import gzip
import os
import random
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def write_bam(path, refs, recs):
    """Minimal BAM writer (several gzip members, like BGZF): recs = (rid, pos0, mapq, flag, [(len, op)], qname, seq_len)."""
    out = [b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", len(refs))]
    for name, ln in refs:
        out.append(struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln))
    for (rid, pos, mapq, flag, cig, qname, l_seq) in recs:
        body = struct.pack("<iiBBHHHiiii", rid, pos, len(qname) + 1, mapq, 4680, len(cig), flag, l_seq, -1, -1, 0)
        body += qname.encode() + b"\0" + b"".join(struct.pack("<I", l << 4 | "MIDNSHP=X".index(o)) for l, o in cig)
        body += b"\x11" * ((l_seq + 1) // 2) + b"\x28" * l_seq
        out.append(struct.pack("<i", len(body)) + body)
    blob = b"".join(out)
    with open(path, "wb") as f:
        for i in range(0, len(blob), 60000):
            f.write(gzip.compress(blob[i:i + 60000]))


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    d = tmp_path_factory.mktemp("n1")
    rng = random.Random(7)
    refs = [("chr1", 200000), ("chr2", 100000)]
    recs = []; names = []
    for i in range(4000):
        nm = "read%d" % i; names.append(nm)
        rid = rng.randrange(2); p1 = rng.randrange(0, refs[rid][1] - 700); p2 = p1 + rng.randrange(150, 400)
        for mate, pos in ((0, p1), (1, p2)):
            r = rng.random()
            cig = [(150, "M")] if r < 0.6 else [(70, "M"), (2, "D"), (80, "M")] if r < 0.75 else [(20, "S"), (130, "M")] if r < 0.9 else [(100, "M"), (3, "I"), (47, "M")]
            mapq = rng.choice([0, 5, 10, 11, 30, 60])
            flag = 1 | (64 if mate == 0 else 128) | (16 if rng.random() < 0.5 else 0)
            if rng.random() < 0.03:
                flag |= 4
            recs.append((rid, pos, mapq, flag, cig, nm, 150))
            if rng.random() < 0.02:      # a secondary alignment of the same read elsewhere
                recs.append((rid, max(0, pos - 5000), 0, flag | 256, [(150, "M")], nm, 150))
    for i in range(200):                 # single-end reads (no mate flags), names that end in 1 or 2 included
        recs.append((0, rng.randrange(0, 199000), rng.choice([0, 60]), 0, [(150, "M")], "se%d" % i, 150))
    recs.sort(key=lambda r: (r[0], r[1]))
    bam = str(d / "reads.bam"); write_bam(bam, refs, recs)
    bed = str(d / "regions.bed")
    with open(bed, "w") as f:
        for _ in range(60):
            c = rng.randrange(2); b = rng.randrange(0, refs[c][1] - 5000); f.write("%s\t%d\t%d\n" % (refs[c][0], b, b + rng.randrange(200, 5000)))
        f.write("chr1\t100\t90000\nchr1\t50000\t150000\n")     # overlapping lines
    fq = []
    for m in (1, 2):
        p = str(d / ("r_%d.fq" % m)); fq.append(p)
        with open(p, "w") as f:
            order = names[:] + ["se%d" % i for i in range(200)] if m == 1 else names[:]
            if m == 2:
                del order[100:160]       # some mates missing from file 2 -> singletons
            for nm in order:
                s = "".join(rng.choice("ACGT") for _ in range(150))
                f.write("@%s extra comment\n%s\n+\n%s\n" % (nm, s, "I" * 150))
    return dict(dir=str(d), bam=bam, bed=bed, fq=fq)


@pytest.mark.parametrize("prune", [True, False])
def test_extract_reads_matches_oracle(case, prune):
    import n1_oracle as o
    exp = o.extract_reads(case["bam"], open(case["bed"]).read().split("\n"), 150, prune=prune)
    cmd = [CLI, "extract-reads"] + ([] if prune else ["--noprune"]) + [case["bam"], case["bed"], "150"]
    r = subprocess.run(cmd, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    got = r.stdout.decode().split("\n")[:-1]
    assert len(got) > 100 and got == exp


def test_extract_sequence_matches_oracle_and_seqtk(case, tmp_path):
    import n1_oracle as o
    rows = o.extract_reads(case["bam"], open(case["bed"]).read().split("\n"), 150, prune=True)
    rows_fn = str(tmp_path / "rows.bed"); open(rows_fn, "w").write("\n".join(rows) + "\n")
    e1, e2, es = o.extract_sequence(case["fq"][0], case["fq"][1], rows)
    r = subprocess.run([CLI, "extract-sequence", case["fq"][0], case["fq"][1], rows_fn, str(tmp_path)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert open(tmp_path / "reads_1.fastq", "rb").read() == e1 and open(tmp_path / "reads_2.fastq", "rb").read() == e2
    assert open(tmp_path / "singletons.fastq", "rb").read() == es and len(es) > 0 and len(e1) > 0
    # the subset step against the vendored seqtk itself: same reads, same order (names / comments differ by the later renaming)
    seqtk = os.path.join(ROOT, "oracle", "_ref", "seqtk")
    if not os.path.exists(seqtk):
        pytest.skip("oracle/_ref/seqtk not built")
    lst = str(tmp_path / "l1.txt")
    open(lst, "w").write("\n".join(sorted({x.split("\t")[3][:-2] for x in rows if x.split("\t")[3][-1] == "1"})) + "\n")
    sub = subprocess.run([seqtk, "subseq", case["fq"][0], lst], capture_output=True, check=True).stdout.split(b"\n")
    seqs_seqtk = sub[1::4]
    mine = sorted(e1.split(b"\n")[1::4] + [s for s in es.split(b"\n")[1::4]])
    assert set(seqs_seqtk) <= set(mine) and len(seqs_seqtk) > 0
    got1 = o.subseq(case["fq"][0], [x.encode() for x in open(lst).read().split()])
    assert [r[2] for r in got1] == seqs_seqtk


def test_fused_extraction_leaves_the_same_fastq_texts_in_memory(case, tmp_path):
    """al_extract_to_memory (N1 fused, no files): the three memory files hold exactly what extract-reads | extract-sequence write."""
    import ctypes as C
    rows = subprocess.run([CLI, "extract-reads", case["bam"], case["bed"], "150"], capture_output=True, check=True).stdout
    rows_fn = str(tmp_path / "rows.bed"); open(rows_fn, "wb").write(rows)
    subprocess.run([CLI, "extract-sequence", case["fq"][0], case["fq"][1], rows_fn, str(tmp_path)], capture_output=True, check=True)
    L = C.CDLL(os.path.join(ROOT, "airlift_amd", "lib", "libairlift.so"))
    L.al_extract_to_memory.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.al_extract_to_memory.restype = C.c_int
    fds = (C.c_int * 3)(); npairs, nsingle = C.c_int64(), C.c_int64()
    assert L.al_extract_to_memory(case["bam"].encode(), case["bed"].encode(), 150, 1, case["fq"][0].encode(), case["fq"][1].encode(), fds, C.byref(npairs), C.byref(nsingle)) == 0
    got = []
    for fd in fds:
        assert fd >= 0
        got.append(open("/proc/self/fd/%d" % fd, "rb").read()); os.close(fd)
    exp = [open(tmp_path / n, "rb").read() for n in ("reads_1.fastq", "reads_2.fastq", "singletons.fastq")]
    assert got == exp and npairs.value == exp[0].count(b"\n") // 4 and nsingle.value == exp[2].count(b"\n") // 4 and npairs.value > 100
    # a BAM cut short is an error, and no descriptor is left behind
    bad = str(tmp_path / "cut.bam"); open(bad, "wb").write(open(case["bam"], "rb").read()[:3000])
    assert L.al_extract_to_memory(bad.encode(), case["bed"].encode(), 150, 1, case["fq"][0].encode(), case["fq"][1].encode(), fds, C.byref(npairs), C.byref(nsingle)) < 0
    assert list(fds) == [-1, -1, -1]
