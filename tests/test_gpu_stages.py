"""-m gpu parity tests, stage level: the HIP kernels K1..K4 (through the C-ABI) against
(a) the oracle's sketch and (b) the golden --print-seeds taps the reference build produced."""
import json
import os

import numpy as np
import pytest

from gpu_util import OracleLib, load_fragments, revcomp, seed_lines, split_expected_seeds

pytestmark = pytest.mark.gpu
SETS = ["g1_mt150pe", "g2_100se", "g2_250pe", "g3_adversarial"]


@pytest.fixture(scope="module")
def A():
    import airlift_amd
    airlift_amd.load()
    return airlift_amd


def _run(A, d):
    m, n_segs, seqs, names, quals = load_fragments(d)
    idx = A.Index(fasta=os.path.join(d, m["ref"]))
    ctx = A.Context(idx)
    ctx.upload(n_segs, seqs, names)
    ctx.run()
    return m, n_segs, seqs, names, idx, ctx


@pytest.mark.parametrize("name", SETS)
def test_sketch_matches_oracle(A, oracle_bin, golden_unpacked, name):
    m, n_segs, seqs, names, idx, ctx = _run(A, golden_unpacked[name])
    orc = OracleLib()
    cnt = ctx.tap("mini_cnt", np.uint32, len(seqs))
    off = ctx.tap("mini_off", np.uint64, len(seqs) + 1)
    mini = ctx.tap("mini", np.uint64, int(off[-1]) * 2).reshape(-1, 2)
    r = 0
    for f, ns in enumerate(n_segs):
        for j in range(ns):
            s = seqs[r] if not (ns == 2 and j == 1) else revcomp(seqs[r])   # FR: mate 2 mapped as its reverse complement (map.c:468)
            exp = orc.sketch(s)
            got = mini[int(off[r]): int(off[r]) + int(cnt[r])]
            assert len(exp) == cnt[r], "read %d: %d minimizers, oracle %d" % (r, cnt[r], len(exp))
            assert np.array_equal(got, exp), "read %d minimizers differ" % r
            r += 1
    ctx.close(); idx.close()


@pytest.mark.parametrize("name", SETS)
def test_seeds_and_chains_match_reference_taps(A, golden_unpacked, name):
    d = golden_unpacked[name]
    m, n_segs, seqs, names, idx, ctx = _run(A, d)
    nf = len(n_segs)
    blocks = split_expected_seeds(open(os.path.join(d, "expected.seeds")).read())
    assert len(blocks) == nf
    na1 = ctx.tap("frag_na_p1", np.uint32, nf); off1 = ctx.tap("a_off_p1", np.uint64, nf + 1); rep1 = ctx.tap("frag_rep_p1", np.int32, nf)
    off = ctx.tap("a_off", np.uint64, nf + 1); nu = ctx.tap("frag_nu", np.uint32, nf)
    st = ctx.stat()
    tot = int(st.n_anchor)
    anchors = ctx.tap("anchors", np.uint64, tot * 2).reshape(-1, 2)
    chained = ctx.tap("chained", np.uint64, tot * 2).reshape(-1, 2)
    u = ctx.tap("u", np.uint64, tot + nf + 1)
    uo = ctx.tap("uo", np.uint32, tot + nf + 1)
    na = ctx.tap("frag_na", np.uint32, nf)
    ref_names = idx.names
    bad = 0
    for f in range(nf):
        exp = blocks[f]
        got = ["RS\t%d" % rep1[f]]
        got += seed_lines("SD", ref_names, anchors[int(off1[f]): int(off1[f]) + int(na1[f])])
        exp_sd = [l for l in exp if not l.startswith("CN\t")]
        assert got == exp_sd, "fragment %d (%s): seeds differ\n got %s\n exp %s" % (f, names[sum(n_segs[:f])], got[:6], exp_sd[:6])
    # chains are compared through mm_gen_regs' ordering in the full-pipeline test; here: same multiset of chains
    for f in range(nf):
        exp_cn = [l.split("\t", 2)[2] for l in blocks[f] if l.startswith("CN\t")]
        uu = u[int(off[f]) + f: int(off[f]) + f + int(nu[f])]
        oo = uo[int(off[f]) + f: int(off[f]) + f + int(nu[f])]          # every chain's anchors lie at uo[] inside the fragment's range of chained[]
        got_chains = []
        for c in range(int(nu[f])):
            n = int(uu[c] & np.uint64(0xffffffff)); k = int(off[f]) + int(oo[c])
            assert int(oo[c]) + n <= int(na[f])
            got_chains.append(tuple(l.split("\t", 2)[2] for l in seed_lines("CN", ref_names, chained[k:k + n], 0)))
        # split expected CN lines into chains (gap column of the first anchor of a chain is 0 and id increments)
        exp_chains, cur, last = [], [], None
        for l in blocks[f]:
            if l.startswith("CN\t"):
                cid = l.split("\t")[1]
                if cid != last and cur:
                    exp_chains.append(tuple(cur)); cur = []
                last = cid; cur.append(l.split("\t", 2)[2])
        if cur:
            exp_chains.append(tuple(cur))
        if sorted(got_chains) != sorted(exp_chains):
            bad += 1
            assert bad < 1, "fragment %d chains differ\n got %s\n exp %s" % (f, got_chains[:2], exp_chains[:2])
    assert st.n_sort_tie_flag == 0
    ctx.close(); idx.close()


@pytest.mark.parametrize("name", SETS)
def test_alser_count(A, golden_unpacked, name):
    """a8: every read mapped as a single segment, anchors -> candidate-location count (map.c:299-312)."""
    d = golden_unpacked[name]
    m, n_segs, seqs, names, quals = load_fragments(d)
    import airlift_amd
    files = [airlift_amd.read_fastx(os.path.join(d, r)) for r in m["reads"]]
    nm, sq, _ = files[-1]
    keep = [i for i in range(len(sq)) if len(sq[i]) > 0]
    idx = A.Index(fasta=os.path.join(d, m["ref"]))
    ctx = A.Context(idx)
    ctx.upload([1] * len(keep), [sq[i] for i in keep], [b"" for _ in keep])
    ctx.run()
    assert ctx.alser_count() == m["count"]
    ctx.close(); idx.close()


def _same_index(A, fasta, n_segs=None, seqs=None, names=None, k=None, w=None):
    host = A.Index(fasta=fasta, k=k, w=w)
    dev = A.Index(fasta=fasta, on_device=0, k=k, w=w)
    assert host.names == dev.names
    assert host.stat() == dev.stat()
    assert np.array_equal(host.positions(), dev.positions()), "occurrence arrays differ"
    if seqs:   # and the table answers the same: anchors of a mapped batch are identical
        taps = []
        for idx in (host, dev):
            ctx = A.Context(idx); ctx.upload(n_segs, seqs, names); ctx.run()
            na = ctx.tap("frag_na", np.uint32, len(n_segs)); off = ctx.tap("a_off", np.uint64, len(n_segs) + 1)
            taps.append((na.copy(), ctx.tap("anchors", np.uint64, int(off[-1]) * 2).copy()))
            ctx.close()
        assert np.array_equal(taps[0][0], taps[1][0]) and np.array_equal(taps[0][1], taps[1][1])
    host.close(); dev.close()


@pytest.mark.parametrize("name", ["g1_mt150pe", "g3_adversarial", "g4_q_inv", "g6_repeats"])
def test_device_built_index_equals_host_index(A, golden_unpacked, name):
    d = golden_unpacked[name]
    m, n_segs, seqs, names, quals = load_fragments(d)
    if max(len(x) for x in seqs) > 512:   # long queries are outside this path: compare the index only
        n_segs = seqs = names = None
    _same_index(A, os.path.join(d, m["ref"]), n_segs, seqs, names)


@pytest.mark.parametrize("name", ["g1_mt150pe", "g6_repeats"])
def test_idx_reader_and_max_occ_on_device_index(A, golden_unpacked, name):
    """al_idx_reader_open/read/eof/close (minimap.h:206-232): one part, then NULL; al_idx_cal_max_occ on the device-resident
    table == the host-built index's value == mm_idx_cal_max_occ of the reference build (index.c:164-185)."""
    import ctypes as C
    import subprocess
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    fasta = os.path.join(d, m["ref"])
    L = A.load()
    L.al_idx_reader_open.restype = C.c_void_p; L.al_idx_reader_read.restype = C.c_void_p
    L.al_idx_cal_max_occ.restype = C.c_int32; L.al_idx_cal_max_occ.argtypes = [C.c_void_p, C.c_float]
    io, mo = A.IdxOpt(), A.MapOpt()
    L.al_set_opt(None, C.byref(io), C.byref(mo)); L.al_set_opt(b"sr", C.byref(io), C.byref(mo))
    r = L.al_idx_reader_open(fasta.encode(), C.byref(io), None)
    assert r and L.al_idx_reader_eof(C.c_void_p(r)) == 0
    mi = L.al_idx_reader_read(C.c_void_p(r), 0)
    assert mi and L.al_idx_reader_eof(C.c_void_p(r)) == 1
    assert L.al_idx_reader_read(C.c_void_p(r), 0) is None
    L.al_idx_reader_close(C.c_void_p(r))
    host = A.Index(fasta=fasta)
    mm2ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "mm2ref")
    for f in (2e-4, 0.01, 0.2):
        v = L.al_idx_cal_max_occ(C.c_void_p(mi), f)
        assert v == host.cal_max_occ(f) and v > 0
        if os.path.exists(mm2ref):
            out = subprocess.run([mm2ref, "--max-occ", repr(f), fasta], capture_output=True, text=True)
            assert out.returncode == 0 and v == int(out.stderr.strip().splitlines()[-1].split("value=")[1])
    mo.mid_occ = -1
    L.al_mapopt_update(C.byref(mo), C.c_void_p(mi))
    assert mo.mid_occ == host.cal_max_occ(2e-4)
    host.close(); L.al_idx_destroy(C.c_void_p(mi))


def test_device_built_index_ragged_contigs(A, tmp_path):
    """Contigs shorter than k, shorter than one window, exactly on segment boundaries, with N runs, lower case and IUPAC codes."""
    rng = np.random.default_rng(99)
    lens = [0, 5, 20, 21, 22, 31, 32, 33, 255, 256, 257, 511, 512, 513, 1000, 4096, 70000]
    fa = tmp_path / "ragged.fa"
    with open(fa, "wb") as f:
        for i, L in enumerate(lens):
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].copy()
            if L >= 256:
                for _ in range(3):   # N runs of assorted lengths, some straddling the 256-base segment boundaries
                    p = int(rng.integers(0, L - 40)); s[p:p + int(rng.integers(1, 40))] = ord("N")
                s[250:262] = np.frombuffer(b"NNNNNNNNNNNN", dtype=np.uint8)[:len(s[250:262])]
                s[100:110] = np.frombuffer(b"acgtRYacgt", dtype=np.uint8)
            if L == 70000:
                s[30000:30400] = ord("A")                      # homopolymer: long runs of tied minimizers
                s[40000:40600] = np.frombuffer(b"AC" * 300, dtype=np.uint8)
            f.write(b">c%d some comment\n" % i)
            for o in range(0, L, 70):
                f.write(s[o:o + 70].tobytes() + b"\n")
    _same_index(A, str(fa))


@pytest.mark.parametrize("k,w", [(20, 10), (28, 12), (14, 5), (22, 32)])
def test_device_built_index_even_k_palindromic_repeats(A, tmp_path, k, w):
    """Even k on the device builder: a k-mer equal to its reverse complement is skipped without moving the window (sketch.c:108), so inside (AT)n, (ACGT)n, (GC)n
    or a long inverted repeat the window keeps entries from far back; a lane's lead-in is counted in iterations that move the window and doubled until there are
    enough.  Repeats shorter and much longer than the 256-base segments and the default lead-in, at contig starts and ends, beside N runs; the host builder
    (checked against the reference's index file in the CPU tests) is the comparison."""
    rng = np.random.default_rng(1234 + k)
    def rnd(n):
        return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].tobytes()
    def revcomp(b):
        return b.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
    contigs = []
    contigs.append(rnd(3000) + b"AT" * 40 + rnd(500) + b"AT" * 700 + rnd(700) + b"ACGT" * 900 + rnd(333) + b"GC" * 2000 + rnd(1000))
    contigs.append(b"AT" * 3000 + rnd(2000) + b"TA" * 5000)                               # a repeat at the contig's start and at its end
    contigs.append(rnd(1000) + b"AT" * 300 + b"NNNNN" + b"AT" * 300 + rnd(40) + b"N" + b"CG" * 600 + rnd(1500))
    stem = rnd(5000); contigs.append(rnd(500) + stem + revcomp(stem) + rnd(500))           # one long inverted repeat: a single palindromic k-mer at its centre
    unit = rnd(k // 2); pal = unit + revcomp(unit)                                         # a palindromic k-mer as the repeat unit
    contigs.append(rnd(800) + pal * 400 + rnd(800))
    contigs.append(b"AT" * 20000)                                                          # nothing but the repeat
    contigs.append(rnd(70000))
    fa = tmp_path / ("even_k%d.fa" % k)
    with open(fa, "wb") as f:
        for i, s in enumerate(contigs):
            f.write(b">p%d\n" % i)
            for o in range(0, len(s), 60):
                f.write(s[o:o + 60] + b"\n")
    _same_index(A, str(fa), k=k, w=w)


@pytest.mark.parametrize("name", SETS)
def test_cli_count_candidates_matches_the_unmodified_fork(golden_unpacked, name):
    """`airlift-align --count-candidates` prints the number the as-shipped fork prints instead of alignments
    (main.c:417; golden value produced by the untouched reference tree, oracle/_ref/mm2count)."""
    import json, subprocess
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "airlift_amd", "bin", "airlift-align")
    r = subprocess.run([cli, "-ax", "sr", "--count-candidates", "-K", "30000", m["ref"], m["reads"][-1]], cwd=d, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == b""
    assert ("Total No. of Mappings before alignment (verification): %d" % m["count"]).encode() in r.stderr, r.stderr.decode()[-500:]


def test_rechain_pass_layout_and_run_to_run_determinism(A):
    """Reads inside a high-copy element: most fragments are re-chained with max_occ (map.c:353-375).  The second-pass anchor
    offsets must grow with the fragment id (the chain list of fragment f lives at a_off[f] + f), and two runs of the same batch
    must give the same chains for every fragment (the re-chain list is built with an atomic append; it is sorted before use)."""
    import hashlib, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_synth as g
    ref = g.make_reference(seed=5, n_contigs=2, total_len=900_000, n_dups=4, family=(1500, 200, 0.0))   # copies above mid_occ = 1000
    r1, r2 = g.simulate_pairs(ref, 3000, 150, seed=9, ins_mean=400, ins_sd=40, ins_lo=150, ins_hi=800)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs, names, n_segs = [], [], []
    for i in range(len(r1)):
        n_segs.append(2); seqs += [lut[r1[i]].tobytes(), lut[r2[i]].tobytes()]; names += [b"r%d" % i, b"r%d" % i]
    idx = A.Index(seqs=[lut[c].tobytes() for _, c in ref], names=[n.encode() for n, _ in ref])
    ctx = A.Context(idx)
    nf = len(n_segs); digests = []
    for it in range(3):
        ctx.upload(n_segs, seqs, names); ctx.run()
        st = ctx.stat(); tot = int(st.n_anchor)
        off = ctx.tap("a_off", np.uint64, nf + 1).astype(np.int64); off1 = ctx.tap("a_off_p1", np.uint64, nf + 1).astype(np.int64)
        na = ctx.tap("frag_na", np.uint32, nf).astype(np.int64); nu = ctx.tap("frag_nu", np.uint32, nf).astype(np.int64)
        chained = ctx.tap("chained", np.uint64, tot * 2).reshape(-1, 2); u = ctx.tap("u", np.uint64, tot + nf + 1); uo = ctx.tap("uo", np.uint32, tot + nf + 1).astype(np.int64)
        assert st.n_rechain > 100, "the workload is meant to exercise the re-chain pass (%d)" % st.n_rechain
        re = np.nonzero(off[:nf] != off1[:nf])[0]                     # re-chained fragments got new offsets
        assert len(re) == st.n_rechain and (np.diff(off[re]) > 0).all(), "second-pass offsets must ascend with the fragment id"
        o = np.argsort(off[:nf], kind="stable")
        assert (off[o][:-1] + na[o][:-1] <= off[o][1:]).all()
        d = []
        for f in range(nf):
            uu = u[off[f] + f: off[f] + f + nu[f]]; oo = uo[off[f] + f: off[f] + f + nu[f]]
            h = hashlib.md5(uu.tobytes())
            for c in range(int(nu[f])):   # every chain's anchors lie at uo[] inside the fragment's range of chained[]
                n = int(uu[c] & np.uint64(0xffffffff)); assert oo[c] + n <= na[f]
                h.update(chained[off[f] + oo[c]: off[f] + oo[c] + n].tobytes())
            d.append(h.hexdigest())
        digests.append(d)
    assert digests[0] == digests[1] == digests[2]
    ctx.close(); idx.close()


@pytest.mark.parametrize("form", ["one_cell_per_lane", "two_cells_per_lane"])
@pytest.mark.parametrize("name", ["g1_mt150pe", "g2_100se", "g2_250pe", "g3_adversarial", "g4_q_inv"])
def test_ksw_calls_match_aln_seq_taps(golden_unpacked, name, form, monkeypatch):
    """Every ksw_extd2_sse call the reference made on a golden set (--print-aln-seq taps, align.c:313-339: target, query, flag ->
    ez->score and CIGAR) replayed through the device DP (al_dbg_ksw: register-resident systolic form up to 22 x 16 target cells, LDS
    rows above): score and CIGAR must be identical call by call."""
    import ctypes as C
    import numpy as np
    import airlift_amd as A
    if form == "two_cells_per_lane":      # AL_DBG bit 20: d_ksw_pk (al_dev_ksw2.h) on every call of up to 352 target bases (read when the context is created)
        monkeypatch.setenv("AL_DBG", str(1 << 20))
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    lines = open(os.path.join(d, "expected.alnseq")).read().split("\n")
    calls = []
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines):
        if not lines[i].startswith("===>"):
            i += 1; continue
        flag = int(lines[i].split("flag=")[1].split(",")[0])
        t, q, res = lines[i + 1], lines[i + 2], lines[i + 3]
        calls.append((flag, t, q, res)); i += 4
    calls = [c for c in calls if 0 < len(c[1]) <= 1024 and 0 < len(c[2]) <= 512]
    assert len(calls) >= 5
    code = np.full(256, 4, dtype=np.uint8)
    for k, ch in enumerate(b"ACGT"):
        code[ch] = k
    blob = bytearray(); jobs = np.zeros((len(calls), 6), dtype=np.int32)
    for j, (flag, t, q, _) in enumerate(calls):
        jobs[j] = (len(blob), len(blob) + len(t), len(t), len(q), flag, 0)
        blob += t.encode() + q.encode()
    seqs = code[np.frombuffer(bytes(blob), dtype=np.uint8)]
    idx = A.Index(fasta=os.path.join(d, m["ref"])); ctx = A.Context(idx)
    out = np.zeros((len(calls), 9), dtype=np.int32); cap = 512; cig = np.zeros((len(calls), cap), dtype=np.uint32)
    rc = A.load().al_dbg_ksw(ctx.h, len(calls), seqs.ctypes.data_as(C.c_void_p), seqs.nbytes, jobs.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p), cap)
    assert rc == 0
    bad = []
    for j, (flag, t, q, res) in enumerate(calls):
        n = int(out[j, 8])
        got = "score=%d, cigar=%s" % (out[j, 0], "".join("%d%s" % (int(c) >> 4, "MIDN"[int(c) & 15]) for c in cig[j, :n]))
        if got != res:
            bad.append((j, flag, len(t), len(q), got, res))
    ctx.close(); idx.close()
    assert not bad, bad[:5]


def test_wavefront_radix_restatement_equals_serial(A, oracle_bin):
    """d_rs_sort_wave (counts and small buckets by the wavefront, permutation by lane 0) leaves every permutation the serial
    restatement leaves -- which tests/test_dev_sort_cpu.py pins to the reference's ksort.h -- and the oracle's o_radix_sort_128x."""
    import ctypes as C
    L = A.load()
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    O = C.CDLL(os.path.join(ROOT, "oracle", "libal_oracle.so"))
    O.o_radix_sort_128x.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(21)
    for it in range(60):
        n = int([65, 66, 100, 129, 300, 1000, 2500, 4200, 8192, 9000, 20000, 65535][it % 12])
        kind = it % 6
        if kind == 0:    x = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
        elif kind == 1:  x = rng.integers(0, 4, size=n, dtype=np.uint64)
        elif kind == 2:  x = rng.integers(0, 50, size=n, dtype=np.uint64) << np.uint64(int(rng.integers(0, 57)))
        elif kind == 3:  x = (rng.integers(20, 300, size=n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 32, size=n, dtype=np.uint64) // np.uint64(1 << 22) * np.uint64(1 << 22)   # chain keys: score | hash, with ties
        elif kind == 4:  x = np.full(n, 12345, dtype=np.uint64)
        else:            x = np.sort(rng.integers(0, 300, size=n, dtype=np.uint64))[::-1].copy()
        x = np.ascontiguousarray(x)
        a = np.zeros(n, dtype=np.uint16); b = np.zeros(n, dtype=np.uint16)
        assert L.al_dbg_rs_sort(0, x.ctypes.data_as(C.c_void_p), n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(a, b), (it, n, kind)
        ref = np.empty((n, 2), dtype=np.uint64); ref[:, 0] = x; ref[:, 1] = np.arange(n, dtype=np.uint64)
        O.o_radix_sort_128x(ref.ctypes.data, ref.ctypes.data + ref.nbytes)
        assert np.array_equal(a.astype(np.uint64), ref[:, 1]), (it, n, kind)
