"""TEST INFRASTRUCTURE ONLY -- plain-Python restatement of AirLift's read-extraction stage, used by tests/ as
the checker of `airlift-align extract-reads` / `extract-sequence`.

PARITY UNPINNED: the stage is bash around samtools, bedops (convert2bed), GNU awk (`and()`), seqtk and BBMap (repair.sh /
rename.sh).  Only seqtk is in /root/reference (dependencies/seqtk-v1.3.tar.gz; built by oracle/Makefile into oracle/_ref/seqtk
and used to pin subseq()); samtools / bedops / gawk / BBMap are not in this image, so the rest follows the scripts' text and
the tools' documented behaviour:

  extract_reads(), extract_reads_noprune():  src/4-extract_reads/extract_reads.sh:8, extract_reads_noprune.sh:7
      per BED line (chrom B E): `samtools view BAM chrom:B-E | convert2bed --input=sam -` = one row per MAPPED record that
      overlaps the 1-based closed interval [B, E]: (chrom, POS-1, POS-1 + reference length of the CIGAR, QNAME, MAPQ, strand,
      FLAG, CIGAR, ...); the awk keeps rows with start >= B-1, end <= E-1 and (pruning script only) MAPQ <= 10 or
      CIGAR != "<READSIZE>M"; prints (chrom, start, end, QNAME + ".1" / ".2" by FLAG 64 / 128, MAPQ, CIGAR);
      `sort -uk4,4`: one row per name, the first in input order, rows in byte order of the name.
  extract_sequence():  src/4-extract_reads/extract_sequence.sh:17-19
      names ending in 1 / 2 (last character) minus their last two characters select reads of FASTQ 1 / FASTQ 2 (seqtk subseq:
      exact match of the first word of the header, file order kept, "@name[ comment]" + sequence + "+" + quality);
      repair.sh pairs the two subsets by name (a trailing /1, /2 or " 1:..."/" 2:..." is not part of the name), reads without
      a mate are singletons; rename.sh numbers each output stream from 0: realigned_<n> (both mates of pair n) and
      realigned_singleton_<n>.  BBMap's output order is restated as: pairs in the order of their first mate in FASTQ 1,
      singletons of FASTQ 1 then of FASTQ 2."""
import gzip
import struct


def read_bam(path):
    """Yields (refname, pos0, mapq, flag, cigar_string, ref_len, qname) for every record of a BAM file (any gzip member layout)."""
    with gzip.open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", data, 4); o = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, o); o += 4
    names = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, o); o += 4
        names.append(data[o:o + l - 1].decode()); o += l + 4
    while o < len(data):
        bs, = struct.unpack_from("<i", data, o); r = o + 4; o += 4 + bs
        rid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", data, r)
        qname = data[r + 32:r + 32 + l_rn - 1].decode()
        cig = struct.unpack_from("<%dI" % n_cig, data, r + 32 + l_rn)
        cs = "".join("%d%s" % (c >> 4, "MIDNSHP=X"[c & 15]) for c in cig) or "*"
        rl = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
        yield (names[rid] if rid >= 0 else "*", pos, mapq, flag, cs, rl, qname)


def extract_reads(bam, bed_lines, read_size, prune=True):
    recs = list(read_bam(bam))
    rs = "%dM" % read_size
    rows, seen = [], set()
    for line in bed_lines:
        f = line.split()
        if len(f) < 3:
            continue
        chrom, B, E = f[0], int(f[1]), int(f[2])
        for (rn, pos, mapq, flag, cs, rl, qn) in recs:
            if rn != chrom or (flag & 4):
                continue
            end = pos + rl
            if not (pos < E and max(end, pos + 1) > B - 1):       # samtools region overlap, 1-based closed [B, E]
                continue
            if pos >= B - 1 and end <= E - 1 and (not prune or mapq <= 10 or cs != rs):
                name = qn + (".1" if flag & 64 else ".2" if flag & 128 else "")
                if name not in seen:
                    seen.add(name); rows.append((chrom, pos, end, name, mapq, cs))
    rows.sort(key=lambda r: r[3].encode())
    return ["%s\t%d\t%d\t%s\t%d\t%s" % r for r in rows]


def _fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines) and lines[i]:
        hdr = lines[i][1:]; sp = hdr.split(None, 1)
        yield sp[0], (sp[1] if len(sp) > 1 else b""), lines[i + 1], lines[i + 3]
        i += 4


def subseq(fq, names):
    """seqtk subseq FQ NAMELIST (seqtk.c:546-620): records whose name is listed, in file order."""
    want = set(names)
    return [(n, c, s, q) for (n, c, s, q) in _fastq(fq) if n in want]


def _pair_key(name, comment):
    if len(name) > 2 and name[-2:] in (b"/1", b"/2"):
        return name[:-2]
    return name


def extract_sequence(fq1, fq2, bed_rows):
    l1 = sorted({r.split("\t")[3][:-2].encode() for r in bed_rows if r.split("\t")[3][-1:] == "1"})
    l2 = sorted({r.split("\t")[3][:-2].encode() for r in bed_rows if r.split("\t")[3][-1:] == "2"})
    s1, s2 = subseq(fq1, l1), subseq(fq2, l2)
    k2 = {}
    for i, r in enumerate(s2):
        k2.setdefault(_pair_key(r[0], r[1]), []).append(i)
    used2 = set(); pairs = []; single = []
    for r in s1:
        lst = k2.get(_pair_key(r[0], r[1]))
        j = None
        while lst:
            c = lst.pop(0)
            if c not in used2:
                j = c; break
        if j is None:
            single.append(r)
        else:
            used2.add(j); pairs.append((r, s2[j]))
    single += [r for i, r in enumerate(s2) if i not in used2]
    fmt = lambda nm, r: b"@" + nm + b"\n" + r[2] + b"\n+\n" + r[3] + b"\n"
    o1 = b"".join(fmt(b"realigned_%d" % i, a) for i, (a, b) in enumerate(pairs))
    o2 = b"".join(fmt(b"realigned_%d" % i, b) for i, (a, b) in enumerate(pairs))
    os_ = b"".join(fmt(b"realigned_singleton_%d" % i, r) for i, r in enumerate(single))
    return o1, o2, os_
