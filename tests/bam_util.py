"""Minimal BAM reader for tests (BGZF = concatenated gzip members; record layout of the SAM specification, section 4)."""
import gzip
import struct

SEQ16 = "=ACMGRSVTWYHKDBN"
CIGOPS = "MIDNSHP=XB"


def read_bam(path_or_bytes):
    raw = path_or_bytes if isinstance(path_or_bytes, bytes) else open(path_or_bytes, "rb").read()
    assert raw[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]), "missing BGZF EOF block"
    # every member must carry the BC extra field with the right block size
    p = 0; n_blocks = 0
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04" and raw[p + 12:p + 14] == b"BC"
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        p += bsize; n_blocks += 1
    assert p == len(raw)
    d = gzip.decompress(raw)
    assert d[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", d, 4); text = d[8:8 + l_text].decode(); o = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, o); o += 4
    refs = []
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", d, o); name = d[o + 4:o + 4 + l - 1].decode(); ln, = struct.unpack_from("<i", d, o + 4 + l); refs.append((name, ln)); o += 8 + l
    recs = []
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o); e = o + 4 + bs
        rid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq, nrid, npos, tlen = struct.unpack_from("<iiBBHHHIiii", d, o + 4)
        q = o + 36
        qname = d[q:q + l_rn - 1].decode(); q += l_rn
        cig = "".join("%d%s" % (c >> 4, CIGOPS[c & 0xf]) for c in struct.unpack_from("<%dI" % n_cig, d, q)) or "*"; q += 4 * n_cig
        sb = d[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join(SEQ16[sb[i >> 1] >> 4 if i % 2 == 0 else sb[i >> 1] & 0xf] for i in range(l_seq)) or "*"
        qb = d[q:q + l_seq]; q += l_seq
        qual = "*" if l_seq == 0 or qb[:1] == b"\xff" else "".join(chr(c + 33) for c in qb)
        tags = []
        while q < e:
            tg = d[q:q + 2].decode(); ty = chr(d[q + 2]); q += 3
            if ty == "i":
                v, = struct.unpack_from("<i", d, q); q += 4; tags.append("%s:i:%d" % (tg, v))
            elif ty == "A":
                tags.append("%s:A:%s" % (tg, chr(d[q]))); q += 1
            elif ty == "f":
                v, = struct.unpack_from("<f", d, q); q += 4; tags.append((tg, "f", v))
            elif ty == "Z":
                z = d.index(b"\0", q); tags.append("%s:Z:%s" % (tg, d[q:z].decode())); q = z + 1
            else:
                raise AssertionError("tag type " + ty)
        recs.append(dict(qname=qname, flag=flag, rid=rid, pos=pos, mapq=mapq, cigar=cig, nrid=nrid, npos=npos, tlen=tlen, seq=seq, qual=qual, tags=tags, bin=_bin))
        o = e
    return text, refs, recs, n_blocks


def sam_fields(line, names):
    """SAM text line -> the same dict (tags: de:f parsed to float)."""
    f = line.rstrip("\n").split("\t")
    rid = -1 if f[2] == "*" else names.index(f[2])
    nrid = -1 if f[6] == "*" else (rid if f[6] == "=" else names.index(f[6]))
    tags = []
    for t in f[11:]:
        tg, ty, v = t.split(":", 2)
        tags.append((tg, "f", float(v)) if ty == "f" else t)
    return dict(qname=f[0], flag=int(f[1]), rid=rid, pos=int(f[3]) - 1, mapq=int(f[4]), cigar=f[5], nrid=nrid, npos=int(f[7]) - 1, tlen=int(f[8]), seq=f[9], qual=f[10], tags=tags)
