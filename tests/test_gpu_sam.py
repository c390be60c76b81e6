"""-m gpu parity tests, end to end: the `airlift-align` drop-in (HIP path behind the C-ABI) must print byte-identical
SAM to what the reference build printed for the golden inputs (CIGAR/POS/MAPQ/flags/tags bit-exact)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
SETS = ["g1_mt150pe", "g2_100se", "g2_250pe", "g3_adversarial", "g4_q2", "g6_repeats", "g4_MT_orang", "g4_q_inv"]   # g4_*: the fork's own test/*.fa (16.5 kb and multi-kb queries)


def _diff_report(got, exp, name):
    g, e = got.split(b"\n"), exp.split(b"\n")
    out = ["%s: %d vs %d lines" % (name, len(g), len(e))]
    nbad = 0
    for i in range(min(len(g), len(e))):
        if g[i] != e[i]:
            nbad += 1
            if nbad <= 8:
                gf, ef = g[i].split(b"\t"), e[i].split(b"\t")
                cols = [j for j in range(min(len(gf), len(ef))) if gf[j] != ef[j]]
                out.append("line %d cols %s\n  got %s\n  exp %s" % (i, cols, b"\t".join(gf[:9] + gf[11:]).decode(), b"\t".join(ef[:9] + ef[11:]).decode()))
    out.append("%d differing lines" % nbad)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "samdiff_%s.txt" % name), "w").write("\n".join(out))
    return "\n".join(out)


@pytest.mark.parametrize("name", SETS)
def test_cli_sam_identical(golden_unpacked, name):
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    cmd = [CLI, "-ax", "sr"]
    if m.get("rg"):
        cmd += ["-R", m["rg"]]
    r = subprocess.run(cmd + [m["ref"]] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert hashlib.md5(exp).hexdigest() == m["sam_md5"]
    assert r.stdout == exp, _diff_report(r.stdout, exp, name)


def test_cli_pg_line_as_the_forks_main(golden_unpacked):
    """Without AL_PG_PLAIN the drop-in prints @PG the way main.c:369 does: VN = the fork's MM_VERSION, CL = its own argv;
    everything else is the golden's bytes.  (The format is pinned against the fork's main() in tests/test_capi_cpu.py.)"""
    d = golden_unpacked["g1_mt150pe"]
    m = json.load(open(os.path.join(d, "meta.json")))
    args = ["-ax", "sr", "-t", "2", "-R", m["rg"], m["ref"]] + m["reads"]
    env = {k: v for k, v in os.environ.items() if k != "AL_PG_PLAIN"}
    r = subprocess.run([CLI] + args, cwd=d, capture_output=True, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    bare = b"@PG\tID:minimap2\tPN:minimap2\n"
    assert bare in exp
    assert r.stdout == exp.replace(bare, b"@PG\tID:minimap2\tPN:minimap2\tVN:2.17-r954-dirty\tCL:minimap2 " + " ".join(args).encode() + b"\n")


@pytest.mark.parametrize("name", ["g2_250pe", "g3_adversarial"])
def test_lds_dp_path_identical(golden_unpacked, name):
    """AL_DBG=128 forces every extension through the LDS-row DP (the path long targets take) instead of the
    register-resident one: both must give the reference's bytes."""
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    cmd = [CLI, "-ax", "sr"] + (["-R", m["rg"]] if m.get("rg") else [])
    r = subprocess.run(cmd + [m["ref"]] + m["reads"], cwd=d, capture_output=True, env=dict(os.environ, AL_DBG="128"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert r.stdout == exp, _diff_report(r.stdout, exp, name + "_lds")


@pytest.mark.parametrize("env", [dict(AL_DBG=str(1 << 27)), dict(AL_TEST_SORT_BLK="65"), dict(AL_TEST_SORT_BIG="65"), dict(AL_TEST_SORT_BLK="65", AL_TEST_SORT_BIG="200"),
                                 dict(AL_TEST_SORT_BIG="65", AL_TEST_BIG_CHUNK="3", AL_BIG_MERGE="0"), dict(AL_TEST_SORT_BIG="65", AL_BIG_MERGE="0"), dict(AL_TEST_SORT_BIG="65", AL_TEST_RUN="64,16"), dict(AL_TEST_SORT_BIG="65", AL_TEST_RUN="128,128", AL_TEST_POISON="170", AL_TEST_GUARD="1"),
                                 dict(AL_ORDER_BLOCK="65"), dict(AL_ORDER_BLOCK="100000"), dict(AL_DP_CONC="0"),
                                 dict(AL_HEAP_OLD="1", AL_TEST_HEAP_WAVE="1"), dict(AL_HEAP_OLD="1"), dict(AL_SPEC_MIN="1"), dict(AL_SPEC_MERGE="0"), dict(AL_CHAIN_WAVE_MAX="0"),
                                 dict(AL_TEST_SEG_BIG="160", AL_DBG=str(1 << 27)), dict(AL_TEST_SEG_BIG="64", AL_TEST_POISON="170", AL_TEST_GUARD="1", AL_DBG=str(1 << 28)),
                                 dict(AL_TEST_TILE_ALL="1"), dict(AL_TEST_TILE_ALL="1", AL_TEST_POISON="170", AL_TEST_GUARD="1"), dict(AL_DBG=str(1 << 28)), dict(AL_TEST_TILE_ALL="1", AL_TEST_TILE_FB="1"),
                                 dict(AL_TEST_TILE_ALL="1", AL_CHAIN_COOP="1"), dict(AL_TEST_TILE_ALL="1", AL_CHAIN_COOP="0"),
                                 dict(AL_PREP_HEAVY="2", AL_FIN_HEAVY="2"), dict(AL_PREP_HEAVY="2", AL_FIN_HEAVY="2", AL_TEST_POISON="170", AL_TEST_GUARD="1"), dict(AL_PREP_HEAVY="0", AL_FIN_HEAVY="0"),
                                 dict(AL_REGS_SPLIT="0"), dict(AL_CHAIN_OVL="0", AL_SIDE_PRIO="1")],
                         ids=["segments_wave_only", "block_sort", "run_merge_sort", "block_and_run_merge_sort", "device_radix_sort_chunks", "device_radix_sort", "run_merge_sort_runs_of_64_tiles_of_16", "run_merge_sort_runs_of_128_poisoned_memory", "chain_order_by_a_block_of_16_wavefronts", "chain_order_by_one_wavefront", "dp_classes_one_after_the_other", "serial_heap_merge_by_wavefront", "serial_heap_merge_by_lanes", "merge_ahead_of_the_rechain_pass_every_candidate", "no_merge_ahead_of_the_rechain_pass", "lds_chain_kernels_for_thin_classes",
                              "cut_and_merge_by_eight_wavefronts_all_fragments", "cut_and_merge_by_eight_wavefronts_poisoned_memory",
                              "tile_kernel_all_fragments", "tile_kernel_all_fragments_poisoned_memory", "segment_kernels_instead_of_tiles", "tile_kernel_hands_every_fragment_back",
                              "deferred_segments_sixteen_lanes_each", "deferred_segments_a_lane_each",
                              "prep_and_finish_a_lane_per_hit_every_fragment", "prep_and_finish_a_lane_per_hit_poisoned_memory", "prep_and_finish_a_lane_per_fragment",
                              "chain_post_sort_and_pass_in_one_kernel", "lane_chaining_on_the_main_stream_high_priority_side_streams"])
@pytest.mark.parametrize("name", ["g1_mt150pe", "g2_250pe", "g3_adversarial", "g6_repeats"])
def test_large_fragment_paths_identical(golden_unpacked, name, env):
    """The kernels that take over for fragments with many anchors -- the tile chaining kernel (AL_TEST_TILE_ALL: every fragment goes through it,
    several small fragments per tile; AL_TEST_TILE_FB: it hands every fragment to its fallback, the compact virtual batch), chaining by segments
    (AL_DBG bit 28: instead of the tile kernel; bit 27: every fragment goes through the segment path and the wavefront kernel), the register-network block sort and the device-wide radix sort of anchors (also cut into chunks of three fragments)
    (thresholds lowered so that ordinary fragments reach them), the serial forms of the exact heap merge (AL_HEAP_OLD: a lane or a wavefront per fragment with the heap in LDS; the default since round 5 is the heap in the
    lanes of a wavefront, k_anchor_heap_lanes) for every fragment with equal-x anchors -- must give the reference's bytes as well.  Round 4: k_ext_prep / k_ext_finish with a wavefront per fragment and a lane per hit for EVERY
    fragment (AL_PREP_HEAVY / AL_FIN_HEAVY = 2 job slots) and for none, chain_post's 257 ... 1024-chain class sorted and passed over in one kernel
    (AL_REGS_SPLIT=0), the stream arrangement switches."""
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    cmd = [CLI, "-ax", "sr"] + (["-R", m["rg"]] if m.get("rg") else [])
    r = subprocess.run(cmd + [m["ref"]] + m["reads"], cwd=d, capture_output=True, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert r.stdout == exp, _diff_report(r.stdout, exp, name + "_" + "_".join(env))
    assert b"GUARD:" not in r.stderr, r.stderr.decode()[-1500:]


@pytest.mark.parametrize("name", ["g1_mt150pe", "g3_adversarial", "g6_repeats"])
def test_no_uninitialised_reads_no_overruns(golden_unpacked, name):
    """The allocator's test switches: every device range filled with 0xAA before use (a kernel reading what nobody wrote would
    change the result: fresh hipMalloc ranges are zero, which hides that) and 4 KB guard zones on either side of every range, checked
    after every batch and when a range is freed (a write past a buffer's end lands in the driver's page padding otherwise)."""
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    cmd = [CLI, "-ax", "sr", "-K", "60000"] + (["-R", m["rg"]] if m.get("rg") else [])
    r = subprocess.run(cmd + [m["ref"]] + m["reads"], cwd=d, capture_output=True, env=dict(os.environ, AL_TEST_POISON="170", AL_TEST_GUARD="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"GUARD" not in r.stderr, r.stderr.decode()[-2000:]
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert r.stdout == exp, _diff_report(r.stdout, exp, name + "_poison")


@pytest.mark.parametrize("devs,extra", [("0,0", []), ("0,0,0", ["-K", "20000"]), ("0,0", ["--bam"]), ("0,0,0", ["--bam", "-K", "20000"]), ("0,0", ["--sorted-bam"])], ids=["2lanes", "3lanes_small_batches", "bam", "bam_3lanes_small_batches", "sorted_bam"])
@pytest.mark.parametrize("to_file", [True, False], ids=["pwrite", "pipe"])
def test_multi_lane_identical(golden_unpacked, tmp_path, devs, extra, to_file):
    """--devices with several lanes (here all on GPU 0: the box has one): every mini-batch is cut into contiguous fragment ranges, one
    per lane; output = the single-lane bytes, whether the lanes pwrite() at exchanged offsets (regular file) or take turns (pipe)."""
    d = golden_unpacked["g3_adversarial"]
    m = json.load(open(os.path.join(d, "meta.json")))
    base = [CLI, "-ax", "sr"] + (["-R", m["rg"]] if m.get("rg") else []) + extra
    one = subprocess.run(base + [m["ref"]] + m["reads"], cwd=d, capture_output=True)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    if not extra or extra[0] == "-K":
        assert one.stdout == open(os.path.join(d, "expected.sam"), "rb").read()
    if to_file:
        o = str(tmp_path / "multi.out")
        r = subprocess.run(base + ["--devices", devs, "-o", o, m["ref"]] + m["reads"], cwd=d, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got = open(o, "rb").read()
    else:
        r = subprocess.run(base + ["--devices", devs, m["ref"]] + m["reads"], cwd=d, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got = r.stdout
    if "--bam" in extra and to_file:   # the lanes deflate their own shares into whole BGZF blocks (lane-local): other block boundaries, the same records
        from bam_util import read_bam
        t1, r1, recs1, _ = read_bam(one.stdout); t2, r2, recs2, nb = read_bam(got)
        assert (t1, r1) == (t2, r2) and recs1 == recs2 and len(recs2) > 1000
        return
    assert got == one.stdout


def test_sorted_bam_spilled_runs_identical(golden_unpacked):
    """--sorted-bam with a memory bound far below the output (--sort-mem: sorted runs spilled to temp files, k-way merge at the end,
    like `samtools sort -m`) gives the bytes of the in-memory sort."""
    d = golden_unpacked["g1_mt150pe"]
    m = json.load(open(os.path.join(d, "meta.json")))
    base = [CLI, "-ax", "sr", "--sorted-bam", "-K", "30000"] + (["-R", m["rg"]] if m.get("rg") else [])
    a = subprocess.run(base + [m["ref"]] + m["reads"], cwd=d, capture_output=True)
    b = subprocess.run(base + ["--sort-mem", "40k", m["ref"]] + m["reads"], cwd=d, capture_output=True, env=dict(os.environ, AL_TIMING="1"))
    assert a.returncode == 0 and b.returncode == 0, b.stderr.decode()[-2000:]
    assert b"spilled runs" in b.stderr
    assert len(a.stdout) > 1000 and a.stdout == b.stdout


def test_yeast100k_digest(tmp_path):
    """G5: 100 k pairs on the 12 Mbp synthetic genome (the bench workload's shape): md5 of the whole SAM must equal the
    digest the reference build produced (tests/golden/g5_yeast100k/meta.json)."""
    import gen_synth
    from conftest import GOLDEN
    m = json.load(open(os.path.join(GOLDEN, "g5_yeast100k", "meta.json")))
    gen_synth.generate(m["config"], str(tmp_path), pairs=m["pairs"], seed=m["seed"])
    r = subprocess.run([CLI, "-ax", "sr", "ref.fa", "reads_1.fq", "reads_2.fq"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert hashlib.md5(r.stdout).hexdigest() == m["sam_md5"]


def test_output_file_with_a_launchers_world_size_of_one(golden_unpacked, tmp_path):
    """`-o FILE` when WORLD_SIZE is in the environment but the run is a single process (torchrun --nproc_per_node=1, or WORLD_SIZE exported
    without --rank / --ranked): the file is written, nothing goes to stdout (main.c:183-190)."""
    d = golden_unpacked["g1_mt150pe"]
    m = json.load(open(os.path.join(d, "meta.json")))
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    for env in (dict(WORLD_SIZE="1"), dict(WORLD_SIZE="1", RANK="0"), dict(WORLD_SIZE="4")):
        out = tmp_path / "o.sam"
        if out.exists():
            out.unlink()
        extra = ["--ranked"] if "RANK" in env else []
        cmd = [CLI, "-ax", "sr"] + extra + (["-R", m["rg"]] if m.get("rg") else []) + ["-o", str(out), m["ref"]] + m["reads"]
        r = subprocess.run(cmd, cwd=d, capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        assert r.stdout == b"" and out.read_bytes() == exp, env


def test_bwa_style_argv(golden_unpacked):
    """B1: `<aligner> mem -R RG -t N REF R1 R2` (src/0-align_reads.sh:13) gives the same records."""
    d = golden_unpacked["g1_mt150pe"]
    m = json.load(open(os.path.join(d, "meta.json")))
    r = subprocess.run([CLI, "mem", "-R", m["rg"], "-t", "4", m["ref"]] + m["reads"], cwd=d, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == open(os.path.join(d, "expected.sam"), "rb").read()


def test_reads_above_the_limit_fail_loudly(tmp_path):
    """Reads beyond the long-read kernel's state tiles (32768 bp) must produce an error, never a silent fallback."""
    import gen_synth as g
    ref = g.make_reference(seed=3, n_contigs=1, total_len=200_000)
    g.write_fasta(str(tmp_path / "ref.fa"), ref)
    lut = b"ACGT"
    with open(tmp_path / "long.fa", "wb") as f:
        f.write(b">toolong\n" + bytes(lut[c] for c in ref[0][1][1000:1000 + 40000]) + b"\n")
    r = subprocess.run([CLI, "-ax", "sr", "ref.fa", "long.fa"], cwd=tmp_path, capture_output=True)
    assert r.returncode != 0
    assert b"exceeds the limit" in r.stderr or b"not supported" in r.stderr


def test_stage3_regions_2_to_20kb(tmp_path):
    """B3 as AirLift ships it: align_gaps.sh:14-15 (`aln` + `samse`) is fed whole regions (extract_fasta_regions_with_bedfile.sh:4-7),
    i.e. sequences of kilobases.  300 regions of 2-20 kb cut from a diverged copy of the reference (1 % substitutions, small indels,
    a few N) through the bwa-style argv; SAM must equal the reference build's for the same files."""
    import numpy as np
    import gen_synth as g
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot -- without it this comparison against the reference would silently not run")
    ref = g.make_reference(seed=17, n_contigs=3, total_len=3_000_000, n_dups=30, dup_len=(500, 4000), dup_div=0.03)
    g.write_fasta(str(tmp_path / "old.fa"), ref)
    rng = np.random.default_rng(5)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(tmp_path / "gaps.fa", "wb") as f:
        for i in range(300):
            ci = int(rng.integers(0, 3)); c = ref[ci][1]
            L = int(rng.integers(2000, 20001)); st = int(rng.integers(0, len(c) - L))
            s = c[st:st + L].copy()
            m = rng.random(L) < 0.01
            s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            out = []; p = 0
            for q in sorted(rng.integers(50, L - 50, size=max(1, L // 1500)).tolist()):     # an indel every ~1.5 kb
                if q <= p:
                    continue
                out.append(s[p:q]); d = int(rng.integers(1, 12))
                if rng.random() < 0.5:
                    p = q + d                                                             # deletion
                else:
                    out.append(rng.integers(0, 4, size=d, dtype=np.uint8)); p = q         # insertion
            out.append(s[p:]); s = np.concatenate(out)
            if i % 17 == 0:
                s[100:103] = 4
            if i % 2:
                s = (np.where(s < 4, 3 - s, 4))[::-1]
            f.write(b">chr%d:%d-%d\n" % (ci + 1, st + 1, st + L) + lut[s].tobytes() + b"\n")
    exp = subprocess.run([ref_bin, "-t", "16", "old.fa", "gaps.fa"], cwd=tmp_path, capture_output=True, check=True).stdout
    subprocess.run([CLI, "aln", "-n", "0.08", "-t", "4", "old.fa", "gaps.fa"], cwd=tmp_path, stdout=open(tmp_path / "x.sai", "wb"), check=True)
    r = subprocess.run([CLI, "samse", "old.fa", "x.sai", "gaps.fa"], cwd=tmp_path, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == exp, _diff_report(r.stdout, exp, "stage3_regions")


def test_map_frag_api(golden_unpacked):
    """al_map_frag (minimap.h:334 analogue) on single fragments equals the batch results."""
    import airlift_amd as A
    import ctypes as C
    from gpu_util import load_fragments
    d = golden_unpacked["g3_adversarial"]
    m, n_segs, seqs, names, quals = load_fragments(d)
    idx = A.Index(fasta=os.path.join(d, m["ref"]))
    ctx = A.Context(idx)
    ctx.upload(n_segs[:50], seqs[:100], names[:100]); ctx.run()
    nb, rb, _ = ctx.fetch()
    L = A.load()
    for f in range(0, 50, 7):
        ql = (C.c_int * 2)(len(seqs[2 * f]), len(seqs[2 * f + 1]))
        sq = (C.c_char_p * 2)(seqs[2 * f], seqs[2 * f + 1])
        nr = (C.c_int * 2)(); rg = (C.POINTER(A.Reg) * 2)()
        L.al_map_frag(idx.h, 2, ql, sq, nr, rg, ctx.h, C.byref(idx.mo), names[2 * f])
        for s in range(2):
            assert nr[s] == nb[2 * f + s]
            for k in range(nr[s]):
                a, b = rg[s][k], rb[2 * f + s][k]
                assert (a.rid, a.rs, a.re, a.qs, a.qe, a.mapq, a.rev, a.n_cigar, a.dp_max) == (b.rid, b.rs, b.re, b.qs, b.qe, b.mapq, b.rev, b.n_cigar, b.dp_max)
    ctx.close(); idx.close()


# ---- input formats of the drop-in's reader (al_pipeline.cpp / al_seqio.h): same records whatever the container ----------
def _golden_pe(golden_unpacked):
    d = golden_unpacked["g1_mt150pe"]
    m = json.load(open(os.path.join(d, "meta.json")))
    return d, m, open(os.path.join(d, "expected.sam"), "rb").read()


def _run(cmd, cwd, **kw):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, **kw)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r


def test_gzip_and_crlf_inputs(golden_unpacked, tmp_path):
    import gzip
    d, m, exp = _golden_pe(golden_unpacked)
    for i, fn in enumerate(m["reads"]):
        raw = open(os.path.join(d, fn), "rb").read()
        with gzip.open(tmp_path / ("r%d.fq.gz" % i), "wb") as f:
            f.write(raw)
        open(tmp_path / ("c%d.fq" % i), "wb").write(raw.replace(b"\n", b"\r\n"))
    ref = os.path.join(d, m["ref"])
    with gzip.open(tmp_path / "ref.fa.gz", "wb") as f:
        f.write(open(ref, "rb").read())
    rg = ["-R", m["rg"]] if m.get("rg") else []
    assert _run([CLI, "-ax", "sr", "-t", "3"] + rg + [str(tmp_path / "ref.fa.gz"), "r0.fq.gz", "r1.fq.gz"], tmp_path).stdout == exp
    assert _run([CLI, "-ax", "sr", "-t", "3"] + rg + [ref, "c0.fq", "c1.fq"], tmp_path).stdout == exp


def test_small_device_batches_and_threads(golden_unpacked):
    """-K 20k bases per device batch (about 66 pairs each): many batches through the reader/mapper/writer pipeline,
    every worker count, same bytes in the same order."""
    d, m, exp = _golden_pe(golden_unpacked)
    rg = ["-R", m["rg"]] if m.get("rg") else []
    for t in ("1", "2", "7"):
        assert _run([CLI, "-ax", "sr", "-K", "20000", "-t", t] + rg + [m["ref"]] + m["reads"], d).stdout == exp


def test_batch_is_halved_when_device_workspaces_do_not_fit(golden_unpacked):
    """al_batch_run reports AL_ERR_NOMEM when a workspace cannot be allocated (reads from high-copy repeats: workspaces grow with
    the seed hits) and gives its buffers back; the file driver then runs the batch as two halves, recursively.  The test hook
    AL_TEST_NOMEM_ABOVE makes every batch above 37 fragments 'not fit': same bytes, in the same order, and the halving is announced."""
    d, m, exp = _golden_pe(golden_unpacked)
    rg = ["-R", m["rg"]] if m.get("rg") else []
    r = _run([CLI, "-ax", "sr"] + rg + [m["ref"]] + m["reads"], d, env=dict(os.environ, AL_TEST_NOMEM_ABOVE="37"))
    assert r.stdout == exp
    assert b"does not fit the device workspaces" in r.stderr
    r = _run([CLI, "-ax", "sr", "-K", "20000", "-t", "3"] + rg + [m["ref"]] + m["reads"], d, env=dict(os.environ, AL_TEST_NOMEM_ABOVE="5"))
    assert r.stdout == exp


def test_full_device_gives_nomem_and_a_halved_batch_not_an_abort(golden_unpacked):
    """HBM headroom (round 6): with the device filled to within 2 GB by a FOREIGN process, the mapper's allocations that would leave the
    HIP runtime less than the margin (AL_HBM_MARGIN_MB; the runtime places queue scratch there while kernels run, and dies with
    HSA_STATUS_ERROR_OUT_OF_RESOURCES when it cannot) are refused as out of memory: al_batch_run returns AL_ERR_NOMEM, the file driver runs
    the batch in halves, the bytes are the reference's.  The process must never end by a signal."""
    d = golden_unpacked["g6_repeats"]
    m = json.load(open(os.path.join(d, "meta.json")))
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    rg = ["-R", m["rg"]] if m.get("rg") else []
    cmd = [CLI, "-ax", "sr"] + rg + [m["ref"]] + m["reads"]
    seen_halved = False
    for keep_mb, margin_mb in ((2048, 1024), (3000, 2048), (2500, 1024)):
        fill = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "hbm_fill.py"), str(keep_mb)], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
        try:
            line = fill.stdout.readline().decode()
            assert line.startswith("ready"), line
            r = subprocess.run(cmd, cwd=d, capture_output=True, env=dict(os.environ, AL_HBM_MARGIN_MB=str(margin_mb)), timeout=300)
        finally:
            fill.stdin.close(); fill.wait(timeout=60)
        assert r.returncode >= 0, "killed by signal %d: %s" % (-r.returncode, r.stderr.decode()[-1500:])       # an abort of the runtime is SIGABRT
        assert b"HSA_STATUS_ERROR" not in r.stderr
        if r.returncode == 0:
            assert r.stdout == exp, _diff_report(r.stdout, exp, "g6_full_device_%d" % keep_mb)
            seen_halved = seen_halved or b"does not fit the device workspaces" in r.stderr
        else:                                                                    # nothing fits at all: a message and a non-zero status, no records
            assert b"failed" in r.stderr or b"out of memory" in r.stderr.lower()
    assert seen_halved, "no setting went through AL_ERR_NOMEM and a halved batch"


def test_interleaved_single_file_and_uneven_files(golden_unpacked, oracle_bin, tmp_path):
    """One file with /1 /2 mates adjacent (frag_mode pairing by name, map.c:580-586) and two files of different length
    (bseq.c: extra records skipped, with the reference's warning): compared with the oracle on the same inputs."""
    import airlift_amd as A
    d, m, _ = _golden_pe(golden_unpacked)
    (n1, s1, q1), (n2, s2, q2) = [A.read_fastx(os.path.join(d, f)) for f in m["reads"]]
    with open(tmp_path / "il.fq", "wb") as f:
        for i in range(300):
            base = n1[i][:-2] if n1[i][-2:] == b"/1" else n1[i]
            f.write(b"@" + base + b"/1\n" + s1[i] + b"\n+\n" + q1[i] + b"\n")
            if i % 7 != 3:   # some singletons in between
                f.write(b"@" + base + b"/2\n" + s2[i] + b"\n+\n" + q2[i] + b"\n")
    ref = os.path.join(d, m["ref"])
    got = _run([CLI, "-ax", "sr", "-K", "9000", ref, "il.fq"], tmp_path).stdout
    exp = _run([oracle_bin, ref, "il.fq"], tmp_path).stdout
    assert got == exp, _diff_report(got, exp, "interleaved")
    with open(tmp_path / "a.fq", "wb") as f:
        for i in range(200):
            f.write(b"@" + n1[i] + b"\n" + s1[i] + b"\n+\n" + q1[i] + b"\n")
    with open(tmp_path / "b.fq", "wb") as f:
        for i in range(150):
            f.write(b"@" + n2[i] + b"\n" + s2[i] + b"\n+\n" + q2[i] + b"\n")
    r = _run([CLI, "-ax", "sr", ref, "a.fq", "b.fq"], tmp_path)
    assert b"different number of records" in r.stderr
    exp = _run([oracle_bin, ref, "a.fq", "b.fq"], tmp_path).stdout
    assert r.stdout == exp, _diff_report(r.stdout, exp, "uneven")


def test_parallel_fastq_parser_matches_serial_reader(tmp_path):
    """Uncompressed four-line FASTQ is cut into 16 MB blocks at record boundaries and parsed on worker threads (-t 16); the general
    kseq-grammar reader (AL_SERIAL_PARSE=1) must give the same SAM -- also when the file turns irregular half way (a multi-line
    record, blank lines, CRLF: the serial reader takes over from that block on) and when the last record lacks its newline."""
    import gen_synth as g
    ref = g.make_reference(seed=7, n_contigs=3, total_len=300_000, n_dups=12, tandem=6)
    g.write_fasta(str(tmp_path / "ref.fa"), ref)
    r1, r2 = g.simulate_pairs(ref, 70000, 150, seed=4)
    g.write_fastq(str(tmp_path / "a_1.fq"), r1); g.write_fastq(str(tmp_path / "a_2.fq"), r2)
    for nm in ("a_1.fq", "a_2.fq"):      # irregular twin: record 40000 split over several lines, blank lines, one CRLF record; no final newline
        lines = open(tmp_path / nm, "rb").read().split(b"\n")[:-1]
        i = 4 * 40000
        lines[i + 1:i + 2] = [lines[i + 1][:70], lines[i + 1][70:]]
        lines[i + 4:i + 5] = [lines[i + 4][:33], lines[i + 4][33:]]
        j = 4 * 50000 + 2
        lines[j] = lines[j] + b"\r"
        lines.insert(4 * 10000, b"")
        open(tmp_path / nm.replace("a_", "b_"), "wb").write(b"\n".join(lines))
    for pre in ("a", "b"):
        cmd = [CLI, "-ax", "sr", "-t", "16", "ref.fa", pre + "_1.fq", pre + "_2.fq"]
        par = _run(cmd, tmp_path).stdout
        ser = _run(cmd, tmp_path, env=dict(os.environ, AL_SERIAL_PARSE="1")).stdout
        assert len(par) > 10_000_000 and par == ser, pre


def test_fasta_reads_single_end(golden_unpacked, oracle_bin, tmp_path):
    """B3 shape (align_gaps.sh:14-15): multi-line FASTA queries, single end, `samse` argv."""
    import airlift_amd as A
    d, m, _ = _golden_pe(golden_unpacked)
    n1, s1, _q = A.read_fastx(os.path.join(d, m["reads"][0]))
    with open(tmp_path / "gaps.fa", "wb") as f:
        for i in range(250):
            f.write(b">" + n1[i] + b" some description\n" + s1[i][:70] + b"\n" + s1[i][70:] + b"\n")
    ref = os.path.join(d, m["ref"])
    got = _run([CLI, "samse", ref, "x.sai", "gaps.fa"], tmp_path).stdout
    exp = _run([oracle_bin, ref, "gaps.fa"], tmp_path).stdout
    assert got == exp, _diff_report(got, exp, "fasta_se")


# ---- BAM output: same records as the SAM text; --sorted-bam = `samtools view -F4 | samtools sort` content -----
def _same_record(b, s):
    for k in ("qname", "flag", "rid", "pos", "mapq", "cigar", "nrid", "npos", "tlen", "seq", "qual"):
        assert b[k] == s[k], (k, b, s)
    assert len(b["tags"]) == len(s["tags"])
    for x, y in zip(b["tags"], s["tags"]):
        if isinstance(x, tuple):
            assert x[:2] == y[:2] and abs(x[2] - y[2]) < 1e-6, (x, y)
        else:
            assert x == y, (x, y)


@pytest.mark.parametrize("name", ["g1_mt150pe", "g3_adversarial", "g6_repeats"])
def test_bam_outputs_match_sam(golden_unpacked, name):
    from bam_util import read_bam, sam_fields
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    rg = ["-R", m["rg"]] if m.get("rg") else []
    sam = open(os.path.join(d, "expected.sam")).read().split("\n")
    hdr = [l for l in sam if l.startswith("@")]; body = [l for l in sam if l and not l.startswith("@")]
    # input order
    r = _run([CLI, "-ax", "sr", "-t", "4", "--bam", "-K", "100000"] + rg + [m["ref"]] + m["reads"], d)
    text, refs, recs, n_blocks = read_bam(r.stdout)
    assert text == "\n".join(hdr) + "\n"
    names = [n for n, _ in refs]
    assert [("@SQ\tSN:%s\tLN:%d" % x) for x in refs] == [l for l in hdr if l.startswith("@SQ")]
    assert len(recs) == len(body)
    exp = [sam_fields(l, names) for l in body]
    for b, s in zip(recs, exp):
        _same_record(b, s)
    # coordinate order, mapped only
    r = _run([CLI, "-ax", "sr", "-t", "3", "--sorted-bam", "-l", "1", "-K", "100000"] + rg + [m["ref"]] + m["reads"], d)
    text2, refs2, recs2, _ = read_bam(r.stdout)
    assert text2 == "@HD\tVN:1.6\tSO:coordinate\n" + text and refs2 == refs
    keep = [s for s in exp if not (s["flag"] & 4)]
    keys = [(b["rid"], b["pos"]) for b in recs2]
    assert keys == sorted(keys) and len(recs2) == len(keep)
    order = sorted(range(len(keep)), key=lambda i: (keep[i]["rid"], keep[i]["pos"]))     # stable, like the device radix sort
    for b, i in zip(recs2, order):
        _same_record(b, keep[i])


# ---- non-default options: the drop-in against the reference build itself (oracle/_ref/mm2ref travels with the snapshot) ------
OPTSETS = [
    ["-g", "300", "-F", "1200", "-r", "50"],
    ["-A", "1", "-B", "4", "-O", "6,26", "-E", "2,1"],
    ["-n", "3", "-m", "40", "-s", "60", "-N", "5", "-p", "0.8"],
    ["-z", "40,20", "--end-bonus", "5", "--max-chain-skip", "3"],
    ["-k", "19", "-w", "9"],
    ["-A", "4", "-B", "4", "-O", "4,24", "-E", "2,1"],      # a + b >= q + e: the closed-form flanks switch themselves off
    ["--score-N", "3", "--seed", "7", "-M", "0.3"],
    ["-k", "27", "-w", "10"],        # round 5: two words per window slot of the read sketch (k > 25)
    ["-k", "28", "-w", "12"],        # ... and an even k (a k-mer can be its own reverse complement and is then skipped, sketch.c:108; round 6: on the device builder too)
    ["-k", "26", "-w", "8"],
]


@pytest.mark.parametrize("name", ["g1_mt150pe", "g3_adversarial", "g6_repeats"])
def test_non_default_options_match_reference(golden_unpacked, name):
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot -- without it this comparison against the reference would silently not run")
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    for opts in OPTSETS:
        exp = _run([ref_bin] + opts + [m["ref"]] + m["reads"], d).stdout
        got = _run([CLI, "-ax", "sr"] + opts + [m["ref"]] + m["reads"], d).stdout
        assert got == exp, " ".join(opts) + "\n" + _diff_report(got, exp, name + "_opts")


@pytest.mark.parametrize("opts", [["-k", "25", "-w", "10"], ["-k", "27", "-w", "10"]], ids=["k25_reads_beyond_the_packed_entrys_position_bits", "k27"])
def test_long_reads_at_large_k_match_reference(golden_unpacked, opts):
    """The 16.5 kb query of the fork's own test set at k = 25 (the packed window entry of the read sketch holds positions below 8192 there) and k = 27: two words per
    window slot (d_sketch_wide), compared with the reference build."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot")
    d = golden_unpacked["g4_MT_orang"]
    m = json.load(open(os.path.join(d, "meta.json")))
    exp = _run([ref_bin] + opts + [m["ref"]] + m["reads"], d).stdout
    got = _run([CLI, "-ax", "sr"] + opts + [m["ref"]] + m["reads"], d).stdout
    assert got == exp, " ".join(opts) + "\n" + _diff_report(got, exp, "orang_" + opts[1])


@pytest.mark.parametrize("name", ["g1_mt150pe", "g6_repeats"])
def test_index_file_written_and_read_like_the_references(golden_unpacked, name, tmp_path):
    """`-d FILE` (mm_idx_dump, index.c:438) on the index built on the GPU, and a prebuilt index in place of the FASTA (mm_idx_load, index.c:476): the reference maps with the file
    this program wrote, this program maps with the file the reference wrote (and with its own) -- all three the golden SAM."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot")
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    rg = ["-R", m["rg"]] if m.get("rg") else []
    mine, theirs = str(tmp_path / "mine.mmi"), str(tmp_path / "theirs.mmi")
    r = subprocess.run([CLI, "-ax", "sr", "-d", mine, m["ref"]], cwd=d, capture_output=True)           # index only
    assert r.returncode == 0 and r.stdout == b"" and os.path.getsize(mine) > 0, r.stderr.decode()[-500:]
    assert _run([ref_bin] + rg + [mine] + m["reads"], d).stdout == exp
    assert _run([ref_bin] + rg + ["--save-index", theirs, m["ref"]] + m["reads"], d).stdout == exp
    assert os.path.getsize(mine) == os.path.getsize(theirs)
    assert _run([CLI, "-ax", "sr"] + rg + [theirs] + m["reads"], d).stdout == exp
    assert _run([CLI, "-ax", "sr"] + rg + [mine] + m["reads"], d).stdout == exp
    # -d together with reads: the index is written and the reads are mapped (main.c:374-401)
    both = str(tmp_path / "both.mmi")
    assert _run([CLI, "-ax", "sr"] + rg + ["-d", both, m["ref"]] + m["reads"], d).stdout == exp and os.path.getsize(both) == os.path.getsize(mine)


def test_250bp_pairs_match_reference(tmp_path):
    """BASELINE config 5 shape (250 bp PE, insert N(550,60)): long flanks, where a banded extension can run off the matrix
    (target >= query + w + 1) -- compared with the reference build on freshly simulated reads."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot -- without it this comparison against the reference would silently not run")
    import gen_synth as g
    ref = g.make_reference(seed=11, n_contigs=4, total_len=2_000_000, n_dups=40)
    g.write_fasta(str(tmp_path / "ref.fa"), ref)
    r1, r2 = g.simulate_pairs(ref, 30000, 250, seed=3, ins_mean=550, ins_sd=60, ins_hi=1000, sub_rate=0.004)
    g.write_fastq(str(tmp_path / "r_1.fq"), r1, "realigned_"); g.write_fastq(str(tmp_path / "r_2.fq"), r2, "realigned_")
    exp = _run([ref_bin, "-t", "8", "ref.fa", "r_1.fq", "r_2.fq"], tmp_path).stdout
    got = _run([CLI, "-ax", "sr", "ref.fa", "r_1.fq", "r_2.fq"], tmp_path).stdout
    assert got == exp, _diff_report(got, exp, "pe250")
    # and with the closed-form flanks disabled the same bytes come out of the DP kernels
    got2 = _run([CLI, "-ax", "sr", "ref.fa", "r_1.fq", "r_2.fq"], tmp_path, env=dict(os.environ, AL_DBG=str(1 << 31))).stdout
    assert got2 == exp
    # round 5: the two-cells-per-lane form is the default for every class from 8 blocks up; the one-cell form of the 32-block class, and of every class
    for env in (dict(AL_DP_PK32="0"), dict(AL_DP_PK="0", AL_DBG=str(1 << 31))):
        got3 = _run([CLI, "-ax", "sr", "ref.fa", "r_1.fq", "r_2.fq"], tmp_path, env=dict(os.environ, **env)).stdout
        assert got3 == exp, str(env)


def test_randomised_parity_sweep_against_reference():
    """tools/fuzz_parity.py: assorted read lengths (30-500 bp), single/paired, error / indel / N rates, repeats, insert sizes and
    option sets (band, z-drop around the closed-form guard, scores, thresholds) against the reference build."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot -- without it this comparison against the reference would silently not run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "45", "20261002"], capture_output=True)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_randomised_parity_sweep_on_repeat_rich_references():
    """The same sweep on references that are 70 % repeats: thousands of anchors per fragment, almost every fragment re-chained
    with max_occ (map.c:353-375), more than 64 chains / hits per read (the reference's radix sort instead of its insertion
    sort, ksort.h:147-151), CIGAR arena growth.  Seed 31 holds the two cases that exposed the second-pass chain-list layout
    and the > 64-entry sort order."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot -- without it this comparison against the reference would silently not run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "70", "31"], capture_output=True, env=dict(os.environ, FUZZ_REPEAT="0.7"))
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_selection_reproduces_the_references_in_place_compaction():
    """mm_select_sub(_multi) compact the hit array in place while they still look a hit's parent up by its old index (hit.c:238-255,
    pe.c:6-43): once more hits are kept than that index, the tests are made against the kept hit that now sits there.  Case 99 of
    sweep 778 (250 bp pairs, 70 % repeats, -n 1 -m 10 -s 10: hundreds of short chains and several primaries per pair) is the one that
    showed k_regs_select testing against the true parent instead."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if not os.path.exists(ref_bin):
        pytest.fail("oracle/_ref/mm2ref is missing: the reference build (oracle/Makefile, target ref) must travel to the GPU box with the snapshot -- without it this comparison against the reference would silently not run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "120", "778"], capture_output=True, env=dict(os.environ, FUZZ_REPEAT="0.7", FUZZ_ONLY="99"))
    assert r.returncode == 0 and b"ok   case 99" in r.stdout, r.stdout.decode()[-3000:]


def test_token_stage_matches_reference_script_and_aligner(golden_unpacked):
    """`airlift-align tokens` = the reference's gaps_to_fasta.py (tiling into read-sized tokens) + single-end
    alignment of the tokens; golden SAM made by that script and the reference build (tests/golden/make_g7_tokens.py).
    The tokens are cut on the GPU from the packed gap sequences."""
    d = golden_unpacked["g7_tokens"]
    m = json.load(open(os.path.join(d, "meta.json")))
    exp = open(os.path.join(d, "expected.sam"), "rb").read()
    assert hashlib.md5(exp).hexdigest() == m["sam_md5"]
    for extra in ([], ["-K", "3000", "-t", "5"]):
        r = _run([CLI, "tokens", "--read-size", str(m["read_size"]), "--skip", str(m["skip"])] + extra + [m["ref"], m["gaps"]], d)
        assert r.stdout == exp, _diff_report(r.stdout, exp, "g7_tokens")
    # token batches that "do not fit" are halved like read batches
    r = _run([CLI, "tokens", "--read-size", str(m["read_size"]), "--skip", str(m["skip"]), m["ref"], m["gaps"]], d, env=dict(os.environ, AL_TEST_NOMEM_ABOVE="11"))
    assert r.stdout == exp and b"does not fit the device workspaces" in r.stderr
