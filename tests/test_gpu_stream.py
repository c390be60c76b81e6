"""-m gpu tests of the stream driver (al_stream.hip / al_stream_pipe.cpp): FASTQ record parsing, 4-bit packing (rows a1, a4 of
SURVEY.md 8a: bseq.c:56-130, sketch.c:9-26, map.c:291-293) and SAM text (row a22: format.c:387-544) as kernels.  Every case
must give the bytes of the golden SAM (printed by the reference build) or of the host driver (AL_HOST_IO=1: host parser + host
formatter, itself pinned by tests/test_gpu_sam.py) on the same input."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")


def _run(cmd, cwd, env=None):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r


def _golden(golden_unpacked, name):
    d = golden_unpacked[name]
    m = json.load(open(os.path.join(d, "meta.json")))
    return d, m, open(os.path.join(d, "expected.sam"), "rb").read(), (["-R", m["rg"]] if m.get("rg") else [])


@pytest.mark.parametrize("name", ["g1_mt150pe", "g2_100se", "g2_250pe", "g3_adversarial", "g6_repeats"])
@pytest.mark.parametrize("env", [dict(AL_BATCH_READS="64", AL_CTXS="1", AL_SLOTS="2", AL_PIECE_MB="1"), dict(AL_BATCH_READS="301", AL_CTXS="3", AL_SLOTS="5"),
                                 dict(AL_PROBE_READS="100", AL_ALLOC_GBS="0.001"), dict(AL_OUT_PIECE_MB="1", AL_NO_PWRITE="1")],
                         ids=["64_reads_per_batch", "301_reads_3_contexts", "probe_sizing", "small_out_pieces_write"])
def test_stream_batches_carry_and_contexts(golden_unpacked, name, env, tmp_path):
    """Many small batches: every batch boundary falls inside the loaded text, so each batch starts with the previous one's carry;
    batches take turns on 1 / 3 mapping contexts; the batch size chosen from two probe batches; SAM text leaving through 1 MB
    pieces.  Output to a file (pwrite by several threads) and to a pipe."""
    d, m, exp, rg = _golden(golden_unpacked, name)
    r = _run([CLI, "-ax", "sr", "-t", "8"] + rg + [m["ref"]] + m["reads"], d, env=dict(env, AL_TIMING="1"))
    assert b"stream pipeline" in r.stderr, "the stream driver did not run"
    assert r.stdout == exp
    out = tmp_path / "o.sam"
    _run([CLI, "-ax", "sr", "-t", "8", "-o", str(out)] + rg + [m["ref"]] + m["reads"], d, env=env)
    assert out.read_bytes() == exp


def test_stream_equals_host_driver_on_irregular_text(golden_unpacked, tmp_path):
    """What the device parser takes and where it stops: CRLF line ends, a comment after the name, `+name` separator lines and a last
    record without its newline are strict four-line FASTQ; a multi-line record, blank lines between records or a quality string of
    another length are not -- the batch ends in front of the first such record and the general (kseq grammar) reader continues at that
    byte.  Same bytes as the host driver in every case, and the hand-over is announced."""
    import airlift_amd as A
    d, m, exp, rg = _golden(golden_unpacked, "g1_mt150pe")
    (n1, s1, q1), (n2, s2, q2) = [A.read_fastx(os.path.join(d, f)) for f in m["reads"]]
    ref = os.path.join(d, m["ref"])

    def write(path, names, seqs, quals, style):
        with open(path, "wb") as f:
            for i in range(len(names)):
                nm, s, q = names[i], seqs[i], quals[i]
                if style == "crlf":
                    f.write(b"@" + nm + b" a comment\r\n" + s + b"\r\n+" + nm + b"\r\n" + q + b"\r\n")
                elif style == "multiline" and i == 40:
                    h = len(s) // 2
                    f.write(b"@" + nm + b"\n" + s[:h] + b"\n" + s[h:] + b"\n+\n" + q[:h] + b"\n" + q[h:] + b"\n")
                elif style == "blank" and i == 77:
                    f.write(b"\n\n@" + nm + b"\n" + s + b"\n+\n" + q + b"\n")
                else:
                    f.write(b"@" + nm + b"\n" + s + b"\n+\n" + q + (b"" if style == "noeol" and i == len(names) - 1 else b"\n"))

    for style, resume in (("crlf", False), ("noeol", False), ("multiline", True), ("blank", True)):
        write(tmp_path / "a.fq", n1[:200], s1[:200], q1[:200], style)
        write(tmp_path / "b.fq", n2[:200], s2[:200], q2[:200], style)
        host = _run([CLI, "-ax", "sr", "-t", "4"] + rg + [ref, "a.fq", "b.fq"], tmp_path, env=dict(AL_HOST_IO="1")).stdout
        for extra in (dict(), dict(AL_BATCH_READS="50")):
            r = _run([CLI, "-ax", "sr", "-t", "4"] + rg + [ref, "a.fq", "b.fq"], tmp_path, env=dict(extra, AL_TIMING="1"))
            assert r.stdout == host, style
            assert (b"general reader takes over" in r.stderr) == resume, (style, r.stderr[-600:])


def test_stream_single_file_fragments(golden_unpacked, oracle_bin, tmp_path):
    """One input file: adjacent records with the same name (a trailing /1 /2 ignored) are the reads of a fragment, runs of three or
    more are cut in twos from the front, the last read of a batch waits for the next batch's first (map.c:580-586, bseq.h:31-36).
    Batches of 7 reads make every position of a pair fall on a batch boundary.  Compared with the CPU oracle."""
    import airlift_amd as A
    d, m, _, _ = _golden(golden_unpacked, "g1_mt150pe")
    (n1, s1, q1), (n2, s2, q2) = [A.read_fastx(os.path.join(d, f)) for f in m["reads"]]
    with open(tmp_path / "il.fq", "wb") as f:
        for i in range(240):
            base = n1[i][:-2] if n1[i][-2:] == b"/1" else n1[i]
            f.write(b"@" + base + b"/1\n" + s1[i] + b"\n+\n" + q1[i] + b"\n")
            if i % 5 != 3:
                f.write(b"@" + base + b"/2\n" + s2[i] + b"\n+\n" + q2[i] + b"\n")
            if i % 50 == 49:      # a run of three equal names
                f.write(b"@" + base + b"\n" + s1[i][::-1] + b"\n+\n" + q1[i] + b"\n")
    ref = os.path.join(d, m["ref"])
    exp = _run([oracle_bin, ref, "il.fq"], tmp_path).stdout
    for env in (dict(AL_BATCH_READS="7"), dict(AL_BATCH_READS="64", AL_CTXS="1"), dict()):
        r = _run([CLI, "-ax", "sr", ref, "il.fq"], tmp_path, env=dict(env, AL_TIMING="1"))
        assert b"stream pipeline" in r.stderr
        assert r.stdout == exp, env


def test_stream_cuts_a_batch_that_does_not_fit(golden_unpacked):
    """AL_ERR_NOMEM from al_batch_run in the stream driver: the batch is mapped as two halves, recursively, its SAM pieces written in
    order (two files: by fragments; one file: at fragment boundaries), later batches are smaller."""
    d, m, exp, rg = _golden(golden_unpacked, "g1_mt150pe")
    r = _run([CLI, "-ax", "sr"] + rg + [m["ref"]] + m["reads"], d, env=dict(AL_TEST_NOMEM_ABOVE="37", AL_TIMING="1"))
    assert b"stream pipeline" in r.stderr and b"does not fit the device workspaces" in r.stderr
    assert r.stdout == exp
    d, m, exp, rg = _golden(golden_unpacked, "g2_100se")
    r = _run([CLI, "-ax", "sr"] + rg + [m["ref"]] + m["reads"], d, env=dict(AL_TEST_NOMEM_ABOVE="50", AL_TIMING="1"))
    assert b"does not fit the device workspaces" in r.stderr
    assert r.stdout == exp


def test_stream_options_and_flags(golden_unpacked):
    """--sam-hit-only, secondary records printed (no NO_PRINT_2ND is not reachable from the CLI: -N / -p change which hits survive),
    no read group, U bases: the device formatter against the host formatter."""
    d, m, exp, rg = _golden(golden_unpacked, "g6_repeats")
    for opts in (["--sam-hit-only"], ["-N", "3", "-p", "0.3"], ["-k", "19", "-w", "9"], ["-n", "1", "-m", "10"]):
        host = _run([CLI, "-ax", "sr"] + opts + [m["ref"]] + m["reads"], d, env=dict(AL_HOST_IO="1")).stdout
        got = _run([CLI, "-ax", "sr"] + opts + [m["ref"]] + m["reads"], d, env=dict(AL_BATCH_READS="500")).stdout
        assert got == host, opts


@pytest.mark.parametrize("batch", [None, "37", "1"])
@pytest.mark.parametrize("world", [2, 3])
def test_one_process_per_gpu_mode_on_one_gpu(golden_unpacked, tmp_path, world, batch):
    """SURVEY.md 8e as processes: --rank r --world R.  (Round 6.)  The ranks count the lines of their byte shares of the two FASTQ files, agree
    on ONE grid of batches of G records (batch k is rank k mod R's), and after one all-gather of the sizes of a round's R batches every rank
    writes its batch's SAM text into the one output file at its offset: no part files, no copy at the end.  Here the ranks share device 0, so
    the exchanges go through files in the rendezvous directory (RCCL needs a GPU per rank); the merged file must be the single-process output
    byte for byte -- with the default grid (a few batches per rank), with 37-record batches (many rounds, a ragged last one) and with one
    record per batch (more rounds than some ranks have batches)."""
    d, m, exp, rg = _golden(golden_unpacked, "g1_mt150pe")
    out = tmp_path / "merged.sam"
    env = dict(os.environ, AL_RUN_ID="t%d%s" % (world, batch or "d"), AL_TIMING="1", AL_RANK_TIMEOUT="120")
    if batch:
        env["AL_RANK_BATCH"] = batch
        if batch == "1":
            d, m, exp, rg = _golden(golden_unpacked, "g3_adversarial")
    ps = [subprocess.Popen([CLI, "-ax", "sr", "-t", "4", "--device", "0", "--rank", str(r), "--world", str(world), "--rendezvous", str(tmp_path), "-o", str(out)] + rg + [m["ref"]] + m["reads"],
                           cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in ps]
    assert all(p.returncode == 0 for p in ps), b"\n".join(o[1][-800:] for o in outs).decode()
    assert out.read_bytes() == exp
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    assert b"bytes at offset" in outs[1][1]


@pytest.mark.parametrize("world", [2, 3])
def test_one_process_per_gpu_mode_writes_bam(golden_unpacked, tmp_path, world):
    """--bam under --rank / --world (round 5): every rank deflates its own records into whole BGZF blocks, rank 0 writes header and records straight
    into the merged file, the others' parts go behind it at the exchanged offsets, the last rank's part ends with the EOF block.  The merged file must
    decode to the header and, record for record, the fields of the single-process SAM (the reference has no BAM writer: bytes are unpinned)."""
    from bam_util import read_bam, sam_fields
    from test_gpu_sam import _same_record
    d, m, exp, rg = _golden(golden_unpacked, "g1_mt150pe")
    out = tmp_path / "merged.bam"
    env = dict(os.environ, AL_RUN_ID="b%d" % world, AL_TIMING="1", AL_RANK_TIMEOUT="120")
    ps = [subprocess.Popen([CLI, "-ax", "sr", "-t", "4", "--device", "0", "--bam", "-K", "100000", "--rank", str(r), "--world", str(world), "--rendezvous", str(tmp_path), "-o", str(out)] + rg + [m["ref"]] + m["reads"],
                           cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in ps]
    assert all(p.returncode == 0 for p in ps), b"\n".join(o[1][-800:] for o in outs).decode()
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    sam = exp.decode().split("\n")
    hdr = [l for l in sam if l.startswith("@")]; body = [l for l in sam if l and not l.startswith("@")]
    text, refs, recs, n_blocks = read_bam(str(out))
    assert text == "\n".join(hdr) + "\n"
    names = [n for n, _ in refs]
    assert len(recs) == len(body) and n_blocks >= world
    for b, l in zip(recs, body):
        _same_record(b, sam_fields(l, names))
    assert out.read_bytes()[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])   # one EOF block, at the end


def test_ranked_mode_takes_rank_and_world_from_the_launcher(golden_unpacked, tmp_path):
    """--ranked: RANK / WORLD_SIZE as torchrun exports them (the default rendezvous directory is the output file's)."""
    d, m, exp, rg = _golden(golden_unpacked, "g2_250pe")
    out = tmp_path / "merged.sam"
    ps = [subprocess.Popen([CLI, "-ax", "sr", "-t", "4", "--device", "0", "--ranked", "-o", str(out)] + rg + [m["ref"]] + m["reads"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, AL_RUN_ID="env2", AL_RANK_TIMEOUT="120", RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0")) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in ps]
    assert all(p.returncode == 0 for p in ps), b"\n".join(o[1][-800:] for o in outs).decode()
    assert out.read_bytes() == exp
    r = subprocess.run([CLI, "-ax", "sr", "--ranked", "-o", str(out), m["ref"]] + m["reads"], cwd=d, capture_output=True, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
    assert r.returncode == 1 and b"--ranked needs RANK and WORLD_SIZE" in r.stderr


def test_a_missing_rank_makes_the_others_fail(golden_unpacked, tmp_path):
    """Rank 1 of 2 never starts: rank 0 gives up at the first exchange after AL_RANK_TIMEOUT seconds, with a message and status 1."""
    d, m, exp, rg = _golden(golden_unpacked, "g1_mt150pe")
    r = subprocess.run([CLI, "-ax", "sr", "--device", "0", "--rank", "0", "--world", "2", "--rendezvous", str(tmp_path), "-o", str(tmp_path / "o.sam")] + rg + [m["ref"]] + m["reads"],
                       cwd=d, capture_output=True, env=dict(os.environ, AL_RUN_ID="lonely", AL_RANK_TIMEOUT="3"), timeout=300)
    assert r.returncode == 1
    assert b"did not arrive" in r.stderr
