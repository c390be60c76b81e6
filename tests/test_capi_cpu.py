"""CPU tests (-m "not gpu"): the C-ABI library loads, exports every symbol include/airlift.h declares, the host-side
logic (options, index build, SAM header, read sharding) is right, and there is no CPU compute path."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def A():
    import airlift_amd
    if not os.path.exists(airlift_amd.lib_path()):
        airlift_amd.build()
    airlift_amd.load()
    return airlift_amd


def test_exports_every_declared_symbol(A):
    hdr = open(os.path.join(ROOT, "include", "airlift.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(al_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    L = A.load()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_sr_preset_matches_reference_constants(A):
    """options.c:105-122 over options.c:13-49 (SURVEY.md 8 preset constants)."""
    L = A.load(); io, mo = A.IdxOpt(), A.MapOpt()
    assert L.al_set_opt(None, C.byref(io), C.byref(mo)) == 0
    assert L.al_set_opt(b"sr", C.byref(io), C.byref(mo)) == 0
    assert (io.k, io.w) == (21, 11)
    got = dict(a=mo.a, b=mo.b, q=mo.q, e=mo.e, q2=mo.q2, e2=mo.e2, zdrop=mo.zdrop, zdrop_inv=mo.zdrop_inv, end_bonus=mo.end_bonus,
               max_frag_len=mo.max_frag_len, max_gap=mo.max_gap, bw=mo.bw, min_cnt=mo.min_cnt, min_chain_score=mo.min_chain_score,
               min_dp_max=mo.min_dp_max, best_n=mo.best_n, mid_occ=mo.mid_occ, max_occ=mo.max_occ, pe_ori=mo.pe_ori, pe_bonus=mo.pe_bonus,
               seed=mo.seed, sc_ambi=mo.sc_ambi, max_chain_skip=mo.max_chain_skip, max_chain_iter=mo.max_chain_iter)
    exp = dict(a=2, b=8, q=12, e=2, q2=24, e2=1, zdrop=100, zdrop_inv=100, end_bonus=10, max_frag_len=800, max_gap=100, bw=100, min_cnt=2,
               min_chain_score=25, min_dp_max=40, best_n=20, mid_occ=1000, max_occ=5000, pe_ori=1, pe_bonus=33, seed=11, sc_ambi=1,
               max_chain_skip=25, max_chain_iter=5000)
    assert got == exp
    assert abs(mo.pri_ratio - 0.5) < 1e-7 and abs(mo.mask_level - 0.5) < 1e-7
    assert L.al_check_opt(C.byref(io), C.byref(mo)) == 0
    assert L.al_set_opt(b"map-ont", C.byref(io), C.byref(mo)) == -1       # not on AirLift's path
    mo.e = 0
    assert L.al_check_opt(C.byref(io), C.byref(mo)) < 0                   # options.c:168-172


def test_k_limit_is_the_references(A):
    """sketch.c:84 allows k <= 28.  Round 5: the read sketch keeps two words per window slot where its packed 64-bit entry cannot hold hash | position | strand
    (k of 26 ... 28), and the index builders take an even k (round 6: the device builder too) -- so the option check accepts what the reference accepts and rejects 29."""
    L = A.load(); io, mo = A.IdxOpt(), A.MapOpt()
    L.al_set_opt(None, C.byref(io), C.byref(mo)); L.al_set_opt(b"sr", C.byref(io), C.byref(mo))
    for k in (25, 26, 27, 28):
        io.k = k
        assert L.al_check_opt(C.byref(io), C.byref(mo)) == 0
    io.k = 29
    assert L.al_check_opt(C.byref(io), C.byref(mo)) < 0
    r = subprocess.run([os.path.join(ROOT, "airlift_amd", "bin", "airlift-align"), "-ax", "sr", "-k", "29", "x.fa", "y.fq"], capture_output=True)
    assert r.returncode != 0 and b"k must be <= 28" in r.stderr


def test_index_matches_oracle_sketch(A, oracle_bin, golden_unpacked):
    """Host index build: same minimizer multiset as the oracle's sketch of every contig."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import OracleLib
    d = golden_unpacked["g2_100se"]
    names, seqs, _ = A.read_fastx(os.path.join(d, "syn.fa"))
    idx = A.Index(fasta=os.path.join(d, "syn.fa"))
    st = idx.stat()
    orc = OracleLib()
    tot, keys = 0, set()
    for s in seqs:
        m = orc.sketch(s)
        tot += len(m); keys.update((m[:, 0] >> np.uint64(8)).tolist())
    assert st["n_pos"] == tot and st["n_keys"] == len(keys) and st["n_bases"] == sum(len(s) for s in seqs)
    assert idx.names == [n.decode() for n in names]
    idx.close()


MM2REF = os.path.join(ROOT, "oracle", "_ref", "mm2ref")


def _ref_max_occ(fa, f):
    r = subprocess.run([MM2REF, "--max-occ", repr(f), fa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return int(r.stderr.strip().splitlines()[-1].split("value=")[1])


@pytest.mark.skipif(not os.path.exists(MM2REF), reason="reference build absent")
@pytest.mark.parametrize("gset", ["g6_repeats", "g1_mt150pe"])
def test_index_file_both_ways_with_the_reference(A, golden_unpacked, gset, tmp_path):
    """mm_idx_dump / mm_idx_load (index.c:438-531) on the host-built index: (1) the reference build writes its index, al_idx_load reads it, al_idx_dump writes it again and
    the reference maps with THAT file -- its SAM must be the golden one; (2) the same with an index built here (al_idx_build -> al_idx_dump -> the reference)."""
    d = golden_unpacked[gset]
    m = json.load(open(os.path.join(d, "meta.json")))
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    L = A.load()
    L.al_idx_load.restype = C.c_void_p; L.al_idx_load.argtypes = [C.c_char_p]
    L.al_idx_dump.restype = C.c_int; L.al_idx_dump.argtypes = [C.c_char_p, C.c_void_p]
    L.al_idx_is_idx.restype = C.c_int64; L.al_idx_is_idx.argtypes = [C.c_char_p]
    L.al_idx_destroy.argtypes = [C.c_void_p]
    rg = ["-R", m["rg"]] if m.get("rg") else []
    ref_mmi, mine_mmi, mine2_mmi = str(tmp_path / "ref.mmi"), str(tmp_path / "mine.mmi"), str(tmp_path / "mine2.mmi")
    r = subprocess.run([MM2REF] + rg + ["--save-index", ref_mmi, m["ref"]] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode == 0 and r.stdout == exp
    assert L.al_idx_is_idx(ref_mmi.encode()) == os.path.getsize(ref_mmi) and L.al_idx_is_idx(os.path.join(d, m["ref"]).encode()) == 0
    mi = L.al_idx_load(ref_mmi.encode())
    assert mi
    assert L.al_idx_dump(mine_mmi.encode(), C.c_void_p(mi)) == 0
    L.al_idx_destroy(C.c_void_p(mi))
    assert os.path.getsize(mine_mmi) == os.path.getsize(ref_mmi)              # same content, a bucket's hash pairs in another order
    r = subprocess.run([MM2REF] + rg + [mine_mmi] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode == 0 and r.stdout == exp, r.stderr.decode()[-500:]
    idx = A.Index(fasta=os.path.join(d, m["ref"]))
    assert L.al_idx_dump(mine2_mmi.encode(), C.c_void_p(idx.h)) == 0
    idx.close()
    assert os.path.getsize(mine2_mmi) == os.path.getsize(ref_mmi)
    r = subprocess.run([MM2REF] + rg + [mine2_mmi] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode == 0 and r.stdout == exp, r.stderr.decode()[-500:]


@pytest.mark.skipif(not os.path.exists(MM2REF), reason="reference build absent")
@pytest.mark.parametrize("gset,fa", [("g6_repeats", "rep.fa"), ("g2_100se", "syn.fa"), ("g1_mt150pe", "MT-human.fa")])
def test_cal_max_occ_and_mapopt_update_match_reference(A, golden_unpacked, gset, fa):
    """al_idx_cal_max_occ == mm_idx_cal_max_occ (index.c:164-185) of the reference build on the host-built index;
    al_mapopt_update only touches mid_occ when it is <= 0 (options.c:51-61)."""
    path = os.path.join(golden_unpacked[gset], fa)
    idx = A.Index(fasta=path)
    for f in (2e-4, 0.01, 0.2):
        assert idx.cal_max_occ(f) == _ref_max_occ(path, f), (gset, f)
    L = A.load()
    io, mo = A.IdxOpt(), A.MapOpt()
    L.al_set_opt(None, C.byref(io), C.byref(mo)); L.al_set_opt(b"sr", C.byref(io), C.byref(mo))
    L.al_mapopt_update(C.byref(mo), C.c_void_p(idx.h))
    assert mo.mid_occ == 1000
    mo.mid_occ = 0
    L.al_mapopt_update(C.byref(mo), C.c_void_p(idx.h))
    assert mo.mid_occ == _ref_max_occ(path, 2e-4)
    idx.close()


def test_idx_reader_open_fails_like_reference(A):
    L = A.load()
    L.al_idx_reader_open.restype = C.c_void_p
    io, mo = A.IdxOpt(), A.MapOpt()
    L.al_set_opt(None, C.byref(io), C.byref(mo))
    assert L.al_idx_reader_open(b"/nonexistent/ref.fa", C.byref(io), None) is None       # minimap.h:206
    assert L.al_idx_reader_eof(None) == 1
    L.al_idx_reader_close(None)


def test_parallel_fasta_loader_equals_block_reader(A, golden_unpacked, tmp_path):
    """The index builders' whole-file loader (al_fasta.cpp: header scan, pieces cut at line ends, count + copy by a thread pool) against
    the block reader (kseq.h grammar) on awkward files; files it must decline (gzip, FASTQ, junk before the first header)."""
    import gzip
    L = A.load()
    rng = np.random.default_rng(5)
    def seq(n):
        return np.frombuffer(b"ACGTacgtNnUuRY", dtype=np.uint8)[rng.integers(0, 14, n)].tobytes()
    big = seq(21_000_003)                                   # several 8 MB pieces, cut at line ends
    cases = {}
    cases["wrapped"] = b"".join(b">c%d comment > with gt\n" % i + b"\n".join(s[o:o + 60] for o in range(0, len(s), 60)) + b"\n" for i, s in enumerate([seq(1000), seq(59), seq(60), seq(61), b"", seq(7)]))
    cases["crlf_blank_noeol"] = b">a\tx y\r\nACGT\r\n\r\nTTuU\r\n\n>b\r\n\r\nGG\rA\r\n>c\n" + seq(130)     # '\r' inside a line stays, last line without newline
    cases["one_line_big"] = b">big\n" + big + b"\n>tail desc\n" + seq(100) + b"\n"
    cases["wrapped_big"] = b">w\n" + b"\n".join(big[o:o + 70] for o in range(0, len(big), 70)) + b"\n>z\nAC\n"
    cases["header_only_at_eof"] = b">x\nACGT\n>y"
    cases["many"] = b"".join(b">s%d\n%s\n" % (i, seq(int(rng.integers(0, 200)))) for i in range(5000))
    for name, data in cases.items():
        fn = tmp_path / (name + ".fa")
        fn.write_bytes(data)
        for nt in (1, 3, 8):
            assert L.al_dbg_fasta_selftest(str(fn).encode(), nt) == 0, (name, nt)
    declined = {"fastq": b"@r\nACGT\n+\nIIII\n", "junk_first": b"\n>a\nACGT\n", "plus_line": b">a\nACGT\n+\nIIII\n", "empty_name": b"> x\nACGT\n>b\nAC\n"}
    for name, data in declined.items():
        fn = tmp_path / (name + ".fa")
        fn.write_bytes(data)
        assert L.al_dbg_fasta_selftest(str(fn).encode(), 4) == 1, name
    gz = tmp_path / "r.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(cases["wrapped"])
    assert L.al_dbg_fasta_selftest(str(gz).encode(), 4) == 1


def test_no_cpu_path(A, golden_unpacked):
    """Without a HIP device the context creation must fail loudly (no fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d = golden_unpacked["g4_q2"]
    idx = A.Index(fasta=os.path.join(d, "t2.fa"))
    with pytest.raises(A.AirliftError):
        A.Context(idx)
    r = subprocess.run([os.path.join(ROOT, "airlift_amd", "bin", "airlift-align"), "-ax", "sr", "t2.fa", "q2.fa"], cwd=d, capture_output=True)
    assert r.returncode != 0 and b"no HIP device" in r.stderr
    idx.close()


def test_sam_header(A, golden_unpacked, tmp_path):
    """@SQ/@RG/@PG lines (format.c:116-135) through the C-ABI."""
    d = golden_unpacked["g1_mt150pe"]
    idx = A.Index(fasta=os.path.join(d, "MT-human.fa"))
    L = A.load()
    libc = C.CDLL(None); libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]; libc.fclose.argtypes = [C.c_void_p]
    p = str(tmp_path / "h.sam").encode()
    fp = libc.fopen(p, b"w")
    rgid = C.create_string_buffer(256)
    L.al_write_sam_hdr.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
    L.al_write_sam_hdr(fp, idx.h, b"@RG\\tID:S1\\tSM:S1\\tPL:illumina\\tLB:S1", rgid)
    libc.fclose(fp)
    exp = [l for l in open(os.path.join(d, "expected.sam")).read().split("\n") if l.startswith("@")]
    assert open(p.decode()).read().split("\n")[:-1] == exp
    assert rgid.value == b"S1"
    idx.close()


def test_pg_line_matches_the_forks_main(A, golden_unpacked, tmp_path):
    """@PG VN:/CL: (main.c:369 -> format.c:126-133): al_set_program_line + al_write_sam_hdr print the line the fork's own
    main() prints for the same argv (oracle/_ref/mm2count = the fork as shipped: header, then its candidate counter)."""
    mm2count = os.path.join(ROOT, "oracle", "_ref", "mm2count")
    if not os.path.exists(mm2count):
        pytest.skip("reference build absent")
    d = golden_unpacked["g1_mt150pe"]
    args = ["-ax", "sr", "-t", "2", "-R", "@RG\\tID:S1\\tSM:S1", "MT-human.fa", "g1_1.fq", "g1_2.fq"]
    r = subprocess.run([mm2count] + args, cwd=d, capture_output=True, text=True)
    exp = [l for l in r.stdout.split("\n") if l.startswith("@")]
    assert exp and exp[-1].startswith("@PG\tID:minimap2\tPN:minimap2\tVN:") and "\tCL:minimap2 -ax sr" in exp[-1]
    ver = subprocess.run([mm2count, "--version"], capture_output=True, text=True).stdout.strip()
    hdr = open(os.path.join(ROOT, "include", "airlift.h")).read()
    assert '#define AL_MM_VERSION "%s"' % ver in hdr
    idx = A.Index(fasta=os.path.join(d, "MT-human.fa"))
    L = A.load()
    argv = (C.c_char_p * (len(args) + 1))(b"airlift-align", *[a.encode() for a in args])
    L.al_set_program_line(ver.encode(), len(args) + 1, argv)
    libc = C.CDLL(None); libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]; libc.fclose.argtypes = [C.c_void_p]
    p = str(tmp_path / "h.sam").encode()
    fp = libc.fopen(p, b"w")
    L.al_write_sam_hdr.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
    L.al_write_sam_hdr(fp, idx.h, args[5].encode(), None)
    libc.fclose(fp)
    L.al_set_program_line(None, 0, None)
    assert open(p.decode()).read().split("\n")[:-1] == exp
    idx.close()


@pytest.mark.parametrize("lanes,offsets", [(1, 0), (2, 1), (2, 0), (5, 1), (8, 1), (8, 0), (1, 2), (2, 2), (3, 2), (8, 2)])
def test_multi_lane_output_order(A, tmp_path, lanes, offsets):
    """Product path of N>1 (al_map_file_frag_multi): lanes finish their blocks in any order, the output file must hold
    header + blocks in (batch, lane) order -- offsets from the all-gather of {records, bytes} + pwrite, or ordered turns.
    offsets == 2: BAM bytes, every lane deflates its own share into whole BGZF blocks and writes them at the exchanged offset (SURVEY 8e:
    "BGZF blocks are rank-local"); the file is a sequence of valid BGZF members ending in the EOF block and inflates to header + blocks."""
    L = A.load()
    L.al_dbg_ordered_out_selftest.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]; L.al_dbg_ordered_out_selftest.restype = C.c_int
    assert L.al_dbg_ordered_out_selftest(str(tmp_path / "o.sam").encode(), lanes, 40, offsets) == 0


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from airlift_amd.shard import frag_range, output_offsets
    lo, hi = frag_range(1001, rank, world)
    n_rec = 2 * (hi - lo) + rank; n_bytes = 300 * n_rec
    q.put((rank, lo, hi) + output_offsets(n_rec, n_bytes, rank, world, device="cpu", dist=dist))
    dist.barrier(); dist.destroy_process_group()


def test_read_sharding_two_ranks_gloo():
    """N>1 path on CPU: contiguous fragment ranges + the one all-gather that yields merged-output offsets."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]; res = sorted(q.get(timeout=120) for _ in ps); [p.join(60) for p in ps]
    (r0, lo0, hi0, ro0, bo0, tr0, tb0), (r1, lo1, hi1, ro1, bo1, tr1, tb1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 500, 500, 1001)
    assert ro0 == 0 and bo0 == 0 and ro1 == 2 * 500 and bo1 == 300 * 1000
    assert tr0 == tr1 == 1000 + 1003 and tb0 == tb1 == 300 * 2003


@pytest.mark.parametrize("seed", [1, 7, 2026])
def test_device_sam_formatter_equals_al_write_sam(A, seed):
    """The SAM record routine the GPU runs (al_dev_sam.h: k_sam_len / k_sam_write; mm_write_sam3, format.c:387-544) compiled for
    the CPU: random fragments through it and through al_write_sam (which the golden SAM files pin to the reference), `de:f:%.4f`
    (format.c:292) against printf for every ratio 1 - m/d up to d = 700 and for random doubles."""
    L = A.load()
    L.al_dbg_sam_selftest.argtypes = [C.c_uint64, C.c_int]; L.al_dbg_sam_selftest.restype = C.c_int
    assert L.al_dbg_sam_selftest(seed, 8000) == 0


def test_rank_ranges_tile_the_input(A, tmp_path):
    """One process per GPU (al_map_file_frag_ranked): the byte ranges the ranks find for themselves -- line counts of every rank's share
    of each file, one exchange -- tile both files, start at records and pair record k of file 1 with record k of file 2, for records of
    varying length, quality lines that begin with '@', more ranks than records.  Threads act as the ranks (file exchange)."""
    import random
    L = A.load(); L.al_dbg_ranked_selftest.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
    rnd = random.Random(5)
    with open(tmp_path / "a.fq", "wb") as fa, open(tmp_path / "b.fq", "wb") as fb:
        for i in range(3000):
            l1, l2 = rnd.randint(20, 300), rnd.randint(20, 300)
            fa.write(b"@r%d/1\n" % i + b"A" * l1 + b"\n+\n" + b"@" * l1 + b"\n")
            fb.write(b"@r%d/2 x\n" % (i * 13) + b"C" * l2 + b"\n+r\n" + b"I" * l2 + b"\n")
    for w in (1, 2, 3, 8, 16):
        assert L.al_dbg_ranked_selftest(str(tmp_path / "a.fq").encode(), str(tmp_path / "b.fq").encode(), w, str(tmp_path).encode()) == 0, w
    assert L.al_dbg_ranked_selftest(str(tmp_path / "b.fq").encode(), b"", 4, str(tmp_path).encode()) == 0
    # an empty file (AirLift's singleton FASTQ often is) and a file shorter than the number of ranks: every rank gets a (possibly empty) range
    open(tmp_path / "e.fq", "wb").close()
    with open(tmp_path / "t.fq", "wb") as ft:
        ft.write(b"@x\nA\n+\nI\n")
    for w in (2, 5, 16):
        assert L.al_dbg_ranked_selftest(str(tmp_path / "e.fq").encode(), b"", w, str(tmp_path).encode()) == 0, ("empty", w)
        assert L.al_dbg_ranked_selftest(str(tmp_path / "t.fq").encode(), b"", w, str(tmp_path).encode()) == 0, ("tiny", w)
        assert L.al_dbg_ranked_selftest(str(tmp_path / "t.fq").encode(), str(tmp_path / "t.fq").encode(), w, str(tmp_path).encode()) == 0, ("tiny pair", w)
