#!/usr/bin/env python3
"""bench.py -- reads/s remapped by the MI355X re-alignment hot path (BASELINE.json metric).

A "step" is one pass of the whole hot path (minimizer sketch -> seed lookup -> anchor sort -> chaining ->
banded SW extension -> MAPQ/pairing -> alignment records) over one batch of synthetic 150 bp paired-end reads that is
already resident in HBM.  Workload at N=1: the configuration BASELINE.json's metric is quoted on -- synthetic 150 bp
paired-end reads against a ~3 Gb reference (3.1 Gbp, 45 % repeat content from 200 families, 5 % N,
mason-like reads; generated here with fixed seeds; it fits one GPU: 23 GB index).  --config c2/c3/c5 select the other
BASELINE configs.  N>1: one process per GPU, reads sharded by rank (every rank holds the full index; weak scaling), one
16-byte RCCL all-gather per step for the merged-output offsets.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # (read when the HIP runtime starts, as airlift-align sets it: a context has nine streams and a kernel that runs for 100 ms on one must not share a hardware queue with the main stream; the default is 4)
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

WORKLOADS = {"c2": "C2: yeast-sized synthetic reference pair (16 contigs, 12.16 Mbp, 150 planted duplications)",
             "c2r": "C2R: the C2 reference plus an interspersed 300 bp element family (4000 copies, 10 % divergence, ~10 % of the sequence)",
             "c3": "C3 (SURVEY 8d): ce11-sized synthetic reference (6 contigs, 100.3 Mbp, 3 % of the sequence in 100-5000 bp repeats, copy number 2-50, 1-5 % divergence), mason-like reads",
             "c4": "C4 (SURVEY 8d): GRCh38-sized synthetic reference (24 contigs, 3.1 Gbp, 5 % N blocks, 45 % repeat content from 200 families: Alu-like 300 bp up to 1e5 copies, L1-like 6 kb up to 1e3 copies, 0-15 % divergence), mason-like reads (SNP 1e-3, indel 2e-4 <= 10 bp, error ramp 0.2 -> 1 %)",
             "c5": "C5 (SURVEY 8d): the C4 reference, 250 bp PE, insert N(550, 60)",
             "c4s": "C4S: the C4 repeat model at 1/10 of the size (310 Mbp; copy numbers scaled with the length)",
             "c3u": "C3U: 100.3 Mbp uniform reference + 3000 planted duplications (round 1)",
             "c4u": "C4U: 3.1 Gbp uniform reference + 20000 planted duplications, 2 % N (round 1)"}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0   # measured device copy bandwidth quoted by the same guide: the second denominator SURVEY 8(d) asks for


def make_workload(config, lo, hi, read_len, seed, ref, ins_mean=None):
    """ASCII reads of fragments [lo, hi) of THE workload (one input, defined by config + seed: fragment i belongs to chunk
    i // 250000, simulated with seed + chunk by tools/gen_synth.py -- mason-like for C3..C5), fragment-major (mate1, mate2, ...).
    A rank simulates only the chunks its range touches; chunks run on host threads."""
    import gen_synth as g
    from concurrent.futures import ThreadPoolExecutor
    tr = bytes.maketrans(bytes(range(5)), b"ACGTN")
    out = np.empty((hi - lo, 2, read_len), dtype=np.uint8)
    chunk = 250_000
    over = dict(ins_mean=ins_mean, ins_sd=max(1, ins_mean // 10)) if ins_mean else {}

    def one(c):
        r1, r2 = g.simulate(config, ref, chunk, seed + c, read_len=read_len, **over)
        a, b = max(lo, c * chunk), min(hi, (c + 1) * chunk)
        for m, r in ((0, r1), (1, r2)):
            out[a - lo:b - lo, m] = np.frombuffer(r[a - c * chunk:b - c * chunk].tobytes().translate(tr), dtype=np.uint8).reshape(b - a, read_len)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(one, range(lo // chunk, (hi + chunk - 1) // chunk)))
    return out


def write_fastq_sample(path, arr, mate, prefix="realigned_"):
    n, _, L = arr.shape
    q = b"I" * L
    with open(path, "wb") as f:
        f.write(b"".join(b"@" + (prefix + str(i)).encode() + b"\n" + arr[i, mate].tobytes() + b"\n+\n" + q + b"\n" for i in range(n)))


def write_fastq_fast(path, arr, mate, first=0, prefix=b"realigned_"):
    """Four-line FASTQ of arr[:, mate] (ASCII bases), names <prefix><first + i>, constant quality 'I': one array per run of
    equally long names instead of a Python loop per read."""
    n, _, L = arr.shape
    with open(path, "wb") as f:
        i = 0
        while i < n:
            d = len(str(first + i)); j = min(n, 10 ** d - first)           # records [i, j) have d-digit numbers
            m = j - i; hl = 1 + len(prefix) + d
            rec = np.empty((m, hl + 1 + L + 3 + L + 1), dtype=np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1:1 + len(prefix)] = np.frombuffer(prefix, dtype=np.uint8)
            num = np.arange(first + i, first + j, dtype=np.int64)
            for k in range(d):
                rec[:, hl - 1 - k] = 48 + (num // 10 ** k) % 10
            rec[:, hl] = 10; rec[:, hl + 1:hl + 1 + L] = arr[i:j, mate]
            rec[:, hl + 1 + L] = 10; rec[:, hl + 2 + L] = ord("+"); rec[:, hl + 3 + L] = 10
            rec[:, hl + 4 + L:hl + 4 + 2 * L] = ord("I"); rec[:, -1] = 10
            f.write(rec.tobytes()); i = j


def file_to_file(tmp, ref_fa, a, arr, ref, resident_value):
    """The number a user of the drop-in gets (first read parsed -> last record written): FASTQ files in memory-backed
    storage -> `airlift-align` -> SAM file, a.f2f_pairs pairs of the same workload (the step's reads first).  Start-up (FASTA load,
    index build on the GPU, contexts) is reported apart; the rate is the stream pipeline's own clock around parse .. write."""
    import hashlib
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tmp
    d = tempfile.mkdtemp(prefix="al_f2f_", dir=shm)
    try:
        n = max(a.f2f_pairs, a.pairs)
        t0 = time.time()
        rest = make_workload(a.config, a.pairs, n, a.read_len, 20261002, ref, a.ins_mean) if n > a.pairs else None
        f1, f2 = os.path.join(d, "r_1.fq"), os.path.join(d, "r_2.fq")
        for m, fn in ((0, f1), (1, f2)):
            write_fastq_fast(fn, arr[:a.pairs], m, 0)
            if rest is not None:
                write_fastq_fast(fn + ".b", rest, m, a.pairs)
                with open(fn, "ab") as fo, open(fn + ".b", "rb") as fi:
                    shutil.copyfileobj(fi, fo, 1 << 26)
                os.unlink(fn + ".b")
        del rest
        t_gen = time.time() - t0
        cli = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
        # the SAM file goes where one file takes bytes fastest: on these boxes the page cache of /tmp (7-9 GB/s from parallel pwrite) rather
        # than tmpfs (2.4-3.1 GB/s; tools/micro/outfile_rate.cpp) -- if it has the room
        out_dir = d
        try:
            st = os.statvfs(tmp)
            if st.f_bavail * st.f_frsize > 6 * n * (2 * a.read_len + 160) and os.stat(tmp).st_dev != os.stat(d).st_dev:
                out_dir = tmp
        except OSError:
            pass
        out = os.path.join(out_dir, "f2f_out.sam"); nt = min(os.cpu_count() or 1, 32)
        runs = []
        for rep in range(2):          # reported as {first run, second run}: the first is what a user's cold run gets, nothing is picked
            time.sleep(a.f2f_sleep)   # (stated in the record) the process before this one -- this script's own contexts, or the first run -- has just given back > 100 GB, which the driver scrubs in the background; a process that starts at once waits for that
            t0 = time.time()
            rc = subprocess.run([cli, "-ax", "sr", "-t", str(nt), "-o", out, os.path.join(tmp, ref_fa), f1, f2], stderr=subprocess.PIPE, env=dict(os.environ, AL_PG_PLAIN="1", AL_TIMING="1"))
            wall = time.time() - t0
            err = rc.stderr.decode(errors="replace")
            if rc.returncode != 0:
                return {"error": "airlift-align exit %d: %s" % (rc.returncode, err[-500:])}
            m = re.search(r"stream pipeline: .*? reads, (\d+) records, ([0-9.]+) MB of SAM in ([0-9.]+) s", err)
            mi = re.search(r"index build ([0-9.]+) s", err); ma = re.search(r"allocation calls of the process so far: device (\d+) calls, ([0-9.]+) GB, ([0-9.]+) s", err)
            mb = re.search(r"-> batches of (\d+) reads \((\d+) context\(s\), (\d+) slots", err)
            mr = re.search(r"device memory reserve: ([0-9.]+) of ([0-9.]+) GB obtained in (\d+) chunk\(s\) of [0-9.]+ GB by the background thread in ([0-9.]+) s.*?peak in use ([0-9.]+) GB, (\d+) ranges served, (\d+) requests passed on", err)
            mx = re.search(r"index arrays ([0-9.]+) GB on the device", err)
            runs.append({"wall_s": wall, "pipeline_s": float(m.group(3)) if m else None, "records": int(m.group(1)) if m else None, "sam_mb": float(m.group(2)) if m else None,
                         "index_build_s": float(mi.group(1)) if mi else None, "device_alloc_gb": float(ma.group(2)) if ma else None, "device_alloc_s": float(ma.group(3)) if ma else None,
                         "batch_reads": int(mb.group(1)) if mb else None, "contexts": int(mb.group(2)) if mb else None, "slots": int(mb.group(3)) if mb else None,
                         "reserve_gb": float(mr.group(1)) if mr else None, "reserve_obtained_in_background_s": float(mr.group(4)) if mr else None, "peak_device_gb_in_use": float(mr.group(5)) if mr else None,
                         "requests_passed_on_to_hipMalloc": int(mr.group(7)) if mr else None, "index_gb": float(mx.group(1)) if mx else None,
                         "driver_lines": [l[10:330] for l in err.splitlines() if l.startswith("[airlift] stream pipeline: ") or l.startswith("[airlift] pipeline lane") or l.startswith("[airlift] index: ")][:4]})
        best = next((r for r in runs if r["pipeline_s"]), None)     # the FIRST run is the headline of this leg
        for r in runs:
            if r["pipeline_s"]:
                r["reads_per_s"] = 2 * n / r["pipeline_s"]; r["frac_of_resident_value"] = r["reads_per_s"] / resident_value if resident_value else None
                if r.get("peak_device_gb_in_use") and r.get("index_gb") and r.get("batch_reads") and r.get("contexts"):   # PEAK bytes in use at once (the reserve's own count), index apart, per read of the batches in flight
                    r["peak_workspace_kb_per_read"] = (r["peak_device_gb_in_use"] - r["index_gb"]) * 1e6 / (r["batch_reads"] * r["contexts"])
        res = {"pairs": n, "reads": 2 * n, "host_threads": nt, "storage": shm, "output_storage": out_dir, "fastq_generation_s": t_gen, "sleep_before_each_run_s": a.f2f_sleep, "runs": runs}
        if best:
            res.update({"reads_per_s": 2 * n / best["pipeline_s"], "pipeline_s": best["pipeline_s"], "startup_s": best["wall_s"] - best["pipeline_s"],
                        "whole_process_reads_per_s": 2 * n / best["wall_s"], "frac_of_resident_value": (2 * n / best["pipeline_s"]) / resident_value if resident_value else None,
                        "note": "FASTQ -> SAM through the drop-in: raw file blocks to HBM, record parsing / 4-bit packing / SAM text by kernels, batches of `batch_reads` on `contexts` mapping contexts; pipeline_s = first block read -> last byte written (the process's own clock), start-up (FASTA load + index build on the GPU + contexts; the run's device memory is obtained by a background thread meanwhile: device_alloc_s is what the pipeline still waited for) apart; reads_per_s = the FIRST (cold) run of the process, the second run is in `runs`"})
        # parity: the first cpu-sample pairs are the reads the CPU comparator mapped: its SAM must be the head of this one
        ref_sam = os.path.join(tmp, "cpu.sam")
        if os.path.exists(ref_sam) and os.path.exists(out):
            sz = os.path.getsize(ref_sam); h1, h2 = hashlib.md5(), hashlib.md5()
            with open(ref_sam, "rb") as fa, open(out, "rb") as fb:
                left = sz
                while left > 0:
                    x = fa.read(min(left, 1 << 24)); y = fb.read(len(x))
                    if not x: break
                    h1.update(x); h2.update(y); left -= len(x)
            res["head_identical_to_cpu_baseline_sam"] = h1.hexdigest() == h2.hexdigest()
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)
        try:
            os.unlink(os.path.join(tmp, "f2f_out.sam"))
        except OSError:
            pass


def cpu_baseline(tmp, ref_fa, arr, n_pairs, read_len):
    """Times the CPU comparator on a bounded sample of the same workload (rank 0, N=1 only), then the drop-in CLI on the
    same files: end-to-end wall clock (FASTA + FASTQ in, SAM out) and byte identity of the two SAM streams.
    The reference build (oracle/_ref/mm2ref) maps the sample once per thread count of a sweep inside ONE process (index
    built once, its time reported apart); the best point is the baseline."""
    import hashlib
    import re
    cores = os.cpu_count() or 1
    write_fastq_sample(os.path.join(tmp, "cb_1.fq"), arr[:n_pairs], 0)
    write_fastq_sample(os.path.join(tmp, "cb_2.fq"), arr[:n_pairs], 1)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    sweep = sorted({t for t in (16, 32, 64, 128) if t <= cores} or {cores})
    k_big = "500M"                            # one more point with a ten times larger mini-batch (the fork's -K): fewer, longer parallel sections
    if os.path.exists(ref_bin):
        kind, cmd = "reference", [ref_bin, "-t", ",".join(list(map(str, sweep)) + ["%d@%s" % (min(64, max(sweep)), k_big)])]
    else:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "al_oracle"], check=True)
        kind, cmd, sweep = "port", [os.path.join(ROOT, "oracle", "al_oracle"), "-t", str(cores)], [cores]

    def md5(path):
        h = hashlib.md5()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    t0 = time.time()
    with open(os.path.join(tmp, "cpu.sam"), "wb") as f:
        r = subprocess.run(cmd + [ref_fa, "cb_1.fq", "cb_2.fq"], cwd=tmp, stdout=f, stderr=subprocess.PIPE, env=dict(os.environ, MM2REF_TIMING="1"), check=True)
    t_all = time.time() - t0
    pts = [(int(m.group(1)), float(m.group(3)), float(m.group(4)), int(m.group(2) or 0)) for m in re.finditer(r"\[mm2ref\] threads=(\d+)(?: K=(\d+))? index_s=([0-9.]+) map_s=([0-9.]+)", r.stderr.decode(errors="replace"))]
    if pts:
        best = min(pts, key=lambda x: x[2]); t_idx = pts[0][1]; t_map = best[2]; used = best[0]
        t_cold = t_idx + pts[-1][2]           # what one cold run at the last sweep point costs (index + mapping)
    else:                                   # the port prints no timing: whole wall
        t_idx, t_map, used, t_cold = 0.0, t_all, cores, t_all
    base = {"value": 2 * n_pairs / t_map, "unit": "reads/s", "cores": used, "kind": kind,
            "sample": "first %d pairs x %d bp of the same workload (%d mini-batches at the preset's 50 Mbases), SAM to a file; sweep threads[@mini-batch bases] %s in one process, best = %d threads%s (%.2f s of mapping); index build %.1f s not included"
                      % (n_pairs, read_len, -(-2 * n_pairs * read_len // 50_000_000), "/".join("%d%s:%.2fs" % (p[0], ("@%dM" % (p[3] // 1000000)) if len(p) > 3 and p[3] not in (0, 50_000_000) else "", p[2]) for p in pts), used,
                         (" at -K %dM" % (best[3] // 1000000)) if pts and len(best) > 3 and best[3] not in (0, 50_000_000) else "", t_map, t_idx),
            "sweep": [{"threads": p[0], "mini_batch_bases": (p[3] if len(p) > 3 and p[3] else 50_000_000), "map_s": p[2], "reads_per_s": 2 * n_pairs / p[2]} for p in pts],
            "host_cores": cores}
    cli = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
    nt = min(cores, 32)
    t0 = time.time()
    with open(os.path.join(tmp, "gpu.sam"), "wb") as f:
        rc = subprocess.run([cli, "-ax", "sr", "-t", str(nt), ref_fa, "cb_1.fq", "cb_2.fq"], cwd=tmp, stdout=f, stderr=subprocess.PIPE,
                            env=dict(os.environ, AL_PG_PLAIN="1", AL_TIMING="1"))    # bare @PG line, as the reference driver (no argv) prints it
    t_cli = time.time() - t0
    err = rc.stderr.decode(errors="replace")
    m_idx = re.search(r"index build ([0-9.]+) s", err); m_pipe = re.search(r"pipeline lane 0 .*total ([0-9.]+) s", err)
    if rc.returncode != 0:
        sys.stderr.write("[bench] airlift-align failed (%d): %s\n" % (rc.returncode, rc.stderr.decode(errors="replace")[-2000:]))
        return base, {"error": "airlift-align exit %d" % rc.returncode, "identical_sam": False}
    e2e = {"wall_s": t_cli, "reads_per_s": 2 * n_pairs / t_cli, "host_threads": nt, "cpu_wall_s": t_cold, "speedup_vs_cpu_wall": t_cold / t_cli,
           "identical_sam": md5(os.path.join(tmp, "gpu.sam")) == md5(os.path.join(tmp, "cpu.sam")),
           # the same process without its start-up: FASTQ parse -> pack -> H2D -> map -> D2H -> SAM text -> file (the reference's counterpart
           # is cpu_baseline: its mapping pass without the index build)
           "index_build_s": float(m_idx.group(1)) if m_idx else None, "pipeline_s": float(m_pipe.group(1)) if m_pipe else None,
           "pipeline_reads_per_s": (2 * n_pairs / float(m_pipe.group(1))) if m_pipe and float(m_pipe.group(1)) > 0 else None,
           "note": "whole process, cold start: FASTA parse + index build on the GPU + FASTQ parse + mapping + SAM text; CPU wall = its index build + one mapping pass at all cores"}
    return base, e2e


def load_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/prof.sh on the GPU box with the same bench command); None when no profile has been taken."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None, None
    t = json.load(open(p))
    e = t.get("kernels", {}).get(kernel)
    return (e["bytes_per_launch"], t.get("source")) if e else (None, t.get("source"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=0, help="(default: 1 M; 500 k for c5, the size its earlier lines were quoted at) fragments per GPU per step (C4: about 1250 seed hits and 155 chains per pair, ~100 bytes of workspace per seed hit: 1 M pairs take ~140 GB next to the 23 GB index; a batch that does not fit is halved)")
    ap.add_argument("--read-len", type=int, default=0, help="default: the config's (150; C5: 250)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=2_000_000)   # >= 12 mini-batches of the preset's 50 Mbases: the reference's three-step pipeline has something to overlap
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--f2f-sleep", type=float, default=5.0, help="seconds to wait before each process of the file-to-file leg (recorded in the output)")
    ap.add_argument("--f2f-pairs", type=int, default=6_250_000, help="pairs of the file-to-file leg (FASTQ files -> airlift-align -> SAM file; rank 0, N=1 only; 0 = skip)")
    ap.add_argument("--ins-mean", type=int, default=0, help="mean insert size override (short inserts make the mates overlap: equal-key anchors)")
    ap.add_argument("--config", default="c4", help="workload of tools/gen_synth.py: c4 (default: the configuration BASELINE.json's metric is quoted on -- 150 bp PE against a human-sized reference; fits one GPU), c5 (250 bp), c3 (100 Mbp), c2 (yeast-sized), c2r, c4s, c3u, c4u")
    ap.add_argument("--test-one-gpu", action="store_true", help="N > 1 on a one-GPU box (validation of the sharded path only): every rank uses device 0, collectives over gloo")
    a = ap.parse_args()
    if a.pairs <= 0:
        a.pairs = 500_000 if a.config == "c5" else 1_000_000        # (c5: 250 bp pairs hold 2100 seed hits each -- 1 M pairs fill the HBM to within what the runtime's queues need for scratch)

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.test_one_gpu:
            local = 0; torch.cuda.set_device(0); dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)
    cdev = torch.device("cpu") if a.test_one_gpu else dev          # where the collectives' tensors live

    import gen_synth as g
    import airlift_amd as A
    L = A.load()
    L.al_batch_upload_flat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int64]
    L.al_batch_upload_flat.restype = C.c_int

    # reference pair stand-in (C2 = 16 contigs, 12.16 Mbp, 150 planted duplications); identical on every rank.
    # The index is built on this rank's GPU from the FASTA (al_idx_build_device), as the CLI does.
    if not a.read_len:
        a.read_len = g.CONFIGS[a.config][1]["read_len"]
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="al_bench_")
    shared = None
    if world > 1:
        # N processes on one node: rank 0 generates the reference ONCE (3.1 Gbp: 7 s of all cores, a 3.1 GB FASTA) into a directory every rank sees,
        # the others load the saved contigs and read the same FASTA -- not N generations and N files at the same time
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        shared = os.path.join(base, "al_bench_shared_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "x")))
        if rank == 0:
            os.makedirs(shared, exist_ok=True)
            ref = g.build_reference(a.config, cache_dir=shared)
            g.write_fasta(os.path.join(shared, "ref.fa.tmp"), ref); os.replace(os.path.join(shared, "ref.fa.tmp"), os.path.join(shared, "ref.fa"))
        dist.barrier()
        if rank != 0:
            ref = g.build_reference(a.config, cache_dir=shared)
        ref_fa_path = os.path.join(shared, "ref.fa")
    else:
        ref = g.build_reference(a.config)
        ref_fa_path = os.path.join(tmp, "ref.fa")
        g.write_fasta(ref_fa_path, ref)
    t_gen = time.time() - t0
    t0 = time.time()
    idx = A.Index(fasta=ref_fa_path, on_device=local if world > 1 else 0)
    t_index = time.time() - t0
    # ONE input of pairs * world fragments; rank r maps the contiguous range frag_range(N, r, world) of it
    from airlift_amd.shard import frag_range, output_offsets
    t0 = time.time()
    f_lo, f_hi = frag_range(a.pairs * world, rank, world)
    arr = make_workload(a.config, f_lo, f_hi, a.read_len, 20261002, ref, a.ins_mean)
    t_reads = time.time() - t0
    ctx = A.Context(idx, device=local if world > 1 else 0)
    L.al_ctx_set_threads(ctx.h, min(32, os.cpu_count() or 1))       # host packing threads (outside the timed region)
    L.al_ctx_set_no_taps.argtypes = [C.c_void_p, C.c_int]; L.al_ctx_set_no_taps.restype = None
    L.al_ctx_set_no_taps(ctx.h, 1)                                  # as the file drivers' contexts: no copy of the sorted anchors kept for the debug taps (16 bytes per seed hit: 19 GB of C5's 500 k-pair batch, which fills the HBM otherwise)
    state = {"f_lo": f_lo, "arr": arr}

    def upload(nf):
        n_segs = (C.c_int * nf)(*([2] * nf)); qlens = (C.c_int * (2 * nf))(*([a.read_len] * (2 * nf)))
        return L.al_batch_upload_flat(ctx.h, nf, n_segs, qlens, state["arr"].ctypes.data_as(C.c_char_p), b"realigned_", state["f_lo"]) == 0

    def all_ok(ok):   # a failure on one rank is every rank's failure: nobody is left waiting in a collective
        if world > 1:
            t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev); dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1
        return ok

    # A batch whose seed hits do not fit the device workspaces (al_batch_run -> AL_ERR_NOMEM) is halved, as the file drivers do --
    # on every rank at once, and the input stays ONE input of pairs * world fragments split contiguously: the ranks' ranges are cut again.
    pairs_asked = a.pairs
    while True:
        t0 = time.time(); ok = upload(a.pairs); t_upload = time.time() - t0
        if not all_ok(ok):
            raise SystemExit("upload failed")
        if all_ok(L.al_batch_run(ctx.h) == 0):
            break
        if a.pairs <= 125_000:
            raise SystemExit("al_batch_run failed")
        a.pairs //= 2
        sys.stderr.write("[bench] batch did not fit / failed: retrying with %d pairs per step\n" % a.pairs)
        f_lo, f_hi = frag_range(a.pairs * world, rank, world)
        state["f_lo"] = f_lo; state["arr"] = arr = make_workload(a.config, f_lo, f_hi, a.read_len, 20261002, ref, a.ins_mean)
    ctx.n_frag, ctx.n_reads = a.pairs, 2 * a.pairs

    merged = {}

    def step():
        ctx.run()
        # merged-output offsets of this rank's block: one all-gather of {n_records, n_bytes} per step (RCCL over xGMI when N > 1)
        st = ctx.stat()
        merged["rec_off"], merged["byte_off"], merged["records"], merged["bytes"] = output_offsets(int(st.n_regs_aln), int(st.bytes_out), rank, world, device=cdev, dist=dist)

    for _ in range(a.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    stage_ms = np.zeros(40); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        st = ctx.stat()
        stage_ms += np.array(list(st.ms_kernel))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # PCIe-inclusive rate of the same batch (never `value`): pack + H2D upload, run, D2H of the flat result block
    L.al_batch_fetch_flat.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]; L.al_batch_fetch_flat.restype = C.c_int
    t0 = time.perf_counter(); n_io = 2
    for _ in range(n_io):
        upload(a.pairs); ctx.run()
        nrec, nby = C.c_uint64(), C.c_uint64()
        if L.al_batch_fetch_flat(ctx.h, C.byref(nrec), C.byref(nby)) != 0:
            raise SystemExit("fetch failed")
    torch.cuda.synchronize()
    dt_io = (time.perf_counter() - t0) / n_io
    st = ctx.stat()
    dist_info = None; na = nu = na_p1 = None
    if rank == 0:   # distribution of the per-fragment sizes that decide which kernels run
        na = ctx.tap("frag_na", np.uint32, a.pairs); nu = ctx.tap("frag_nu", np.uint32, a.pairs)
        na_p1 = ctx.tap("frag_na_p1", np.uint32, a.pairs)
        pc = lambda v: {"p50": int(np.percentile(v, 50)), "p90": int(np.percentile(v, 90)), "p99": int(np.percentile(v, 99)), "max": int(v.max())} if len(v) else {}
        dist_info = {"anchors_per_fragment": pc(na), "chains_per_fragment": pc(nu)}
    ctx.close(); idx.close()          # the CPU baseline / CLI legs below start their own processes on this GPU
    if rank == 0:
        names = [L.al_stage_name(i).decode() for i in range(st.n_stage)]
        per = {names[i]: float(stage_ms[i] / a.steps) for i in range(st.n_stage)}
        kern = {names[i]: L.al_stage_kernel(i).decode() for i in range(st.n_stage)}
        # ---- roofline: every stage against ITS OWN algorithmic bytes; the headline fraction is the pipeline's ----
        M, An, Wb = float(st.n_mini), float(st.n_anchor), float(st.n_refbases)
        A1 = float(na_p1.sum()) if na_p1 is not None else An                  # anchors of the first pass (the re-chain pass makes the rest)
        a1 = na_p1 if na_p1 is not None else na
        cls = lambda lo_, hi_: float(a1[(a1 >= lo_) & (a1 <= hi_)].sum())
        groups = [   # (name, intervals, algorithmic bytes)
            ("sketch", ["sketch"], float(st.bytes_in) + 16.0 * M),
            ("seed_lookup", ["seed_lookup", "scan", "size_order"], 16.0 * M),
            ("anchor_sort", ["anchor_sort_small", "anchor_sort", "anchor_sort_blk", "anchor_sort_big", "anchor_heap"], 24.0 * A1),
            ("chain", ["chain_lds32", "chain_lds48", "chain_lds64", "chain_lds128", "chain_tile", "chain_deferred", "chain_fallback", "chain_ties"], 16.0 * A1),
            ("rechain (max_occ pass: seed + sort + chain)", ["rechain"], 56.0 * (An - A1)),
            ("regs (chain_post / seg_gen: no bytes in the contract)", ["regs"], 0.0),
            ("extension", ["ext_prep", "ext_sort", "ext_dp_lane", "ext_dp_g4", "ext_dp_g8", "ext_dp_g12", "ext_dp_g16", "ext_dp_g22", "ext_finish", "compact"], 0.5 * Wb + float(st.bytes_out)),
        ]
        own = {"sketch": float(st.bytes_in) + 16.0 * M, "seed_lookup": 16.0 * M, "anchor_sort_small": 24.0 * cls(0, 64), "anchor_sort": 24.0 * cls(65, 1024),
               "anchor_sort_blk": 24.0 * cls(1025, 8192)}   # (chain_lds32..128: empty intervals, their kernels run on a stream of their own beside the sorts)
        # extension DP kernels: reference windows of their jobs at 4 bits / base + one 48-byte ExtOut record per job (the W/2 + B_out terms)
        for iv, ci in (("ext_dp_g4", 5), ("ext_dp_g8", 6)):
            own[iv] = 0.5 * float(st.dp_target_bases[ci]) + 48.0 * float(st.dp_jobs[ci])
        # (the 9 ... 22-block job class runs as three kernels -- 12, 16, 22 register blocks; its reference windows and records are counted for the class:
        #  the share of each kernel is taken in proportion to its time)
        g7 = [iv for iv in ("ext_dp_g12", "ext_dp_g16", "ext_dp_g22") if per.get(iv, 0) > 0]
        for iv in g7:
            own[iv] = (0.5 * float(st.dp_target_bases[7]) + 48.0 * float(st.dp_jobs[7])) * per[iv] / sum(per[i] for i in g7)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))) if os.path.exists(os.path.join(ROOT, "profiles", "traffic.json")) else {}
        # the DP kernel behind an interval as the committed profile names it: which instance runs depends on the read length (query arrays of 256 or 512 bases)
        # and on the scores (two cells per lane, `true`, or the one-cell form)
        dp_geo = {"ext_dp_g4": (4, 64), "ext_dp_g8": (8, 128), "ext_dp_g12": (12, 192), "ext_dp_g16": (16, 256), "ext_dp_g22": (22, 352)}
        def dp_name(iv):
            nb, t = dp_geo[iv]
            cand = ["k_ext_dp<%d, %d, %d, true>" % (nb, 256 if a.read_len <= 256 else 512, t), "k_ext_dp<%d, 512, %d, true>" % (nb, t), "k_ext_dp<%d, 512, %d, false>" % (nb, t), "k_ext_dp<%d, 512, %d>" % (nb, t)]
            for c_ in cand:
                if c_ in tj.get("kernels", {}):
                    return c_
            return cand[0] if nb > 4 else cand[2]
        for iv in dp_geo:
            if kern.get(iv):
                kern[iv] = dp_name(iv)
        stages = []
        for name, ivs, by in groups:
            ms = sum(per.get(i, 0.0) for i in ivs)
            stages.append({"stage": name, "ms": ms, "algorithmic_bytes": by, "GBps": (by / (ms * 1e-3) / 1e9) if ms > 0 else None,
                           "frac": (by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else None})
        step_ms = sum(per.values())
        alg = float(st.algorithmic_bytes)
        dom = max((k for k in per if kern[k]), key=lambda k: per[k])          # longest interval that is exactly one kernel
        dom_bytes = own.get(dom)
        dom_traffic = (tj.get("kernels", {}).get(kern[dom]) or {}).get("bytes_per_launch")
        roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "peak_measured_copy": HBM_COPY_GBS,
                "achieved": alg / (step_ms * 1e-3) / 1e9, "frac": alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_of_measured_copy": alg / (step_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                "scope": "whole pipeline: algorithmic bytes of one step / sum of the stage intervals (HIP events on the context's stream)",
                "algorithmic_bytes_per_step": alg, "algorithmic_bytes_per_read": alg / (2.0 * a.pairs),
                "traffic": None, "traffic_from_committed_profile": tj.get("total_bytes_per_step") if tj.get("workload") == a.config else None, "traffic_source": tj.get("source"),
                "dominant_kernel": {"kernel": kern[dom], "interval": dom, "ms": per[dom], "algorithmic_bytes": dom_bytes,
                                    "achieved": (dom_bytes / (per[dom] * 1e-3) / 1e9) if dom_bytes else None, "frac": (dom_bytes / (per[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom_bytes else None,
                                    "traffic": None, "traffic_from_committed_profile": dom_traffic if tj.get("workload") == a.config else None,
                                    "jobs": None,
                                    "note": ("integer-VALU bound (see roofline.valu): affine-gap DP cells of its jobs; its HBM bytes are the 4-bit reference windows and the result records (the 9 ... 22-block class has %.0f target bases in %d jobs, over its three kernels)" % (float(st.dp_target_bases[7]), int(st.dp_jobs[7]))) if dom.startswith("ext_dp_g") else (None if dom_bytes is not None else "bytes of this kernel's share of the stage are not separable: see its stage row")},
                "stages": stages}
        # The extension DP is bound by integer VALU issue, not by HBM: its row against THAT ceiling.  Wave-instructions per launch come from the
        # committed SQ_INSTS_VALU pass (profiles/traffic.json, tools/prof.sh).  The ceiling is PER KERNEL (round 6): a wave-instruction does not cost a SIMD a
        # fixed 4 cycles -- profiles/r05_valu_issue.txt measures 2.3 - 2.5 cycles for plain 32-bit VALU instructions, 4.2 - 4.6 for v_pk_*16 and DPP-modified
        # ones -- so it follows from the instruction mix of the kernel's row loop (profiles/valu_mix.json, tools/valu_mix.py: counts by kind x those cycles):
        # 1024 SIMDs x 2.4 GHz / (mean cycles per VALU instruction of that mix).  A kernel absent from the mix file is priced at the plain rate (the highest ceiling).
        try:
            vmix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
        except (OSError, ValueError):
            vmix = {}
        valu = []
        dpk = {iv: dp_name(iv) for iv in dp_geo}
        for iv in ("ext_dp_g4", "ext_dp_g8", "ext_dp_g12", "ext_dp_g16", "ext_dp_g22"):
            ins = (tj.get("kernels", {}).get(dpk[iv]) or {}).get("valu_insts_per_launch") if tj.get("workload") == a.config else None
            ci = {"ext_dp_g4": 5, "ext_dp_g8": 6, "ext_dp_g12": 7, "ext_dp_g16": 7, "ext_dp_g22": 7}[iv]
            own_iv = per.get(iv, 0) > 0.05                                      # (a thin 22-block class runs on a side stream beside the 12- and 16-block kernels: its interval is empty)
            km = (vmix.get("kernels", {}) or {}).get(dpk[iv]) or {}
            cyc = km.get("cycles_per_valu_inst") or (vmix.get("cycles_per_kind", {}) or {}).get("plain", 2.5)
            ceil_k = 1024 * 2.4e9 / cyc
            valu.append({"interval": iv, "kernel": dpk[iv], "ms": per.get(iv), "jobs": int(st.dp_jobs[ci]), "target_bases": int(st.dp_target_bases[ci]),
                         "valu_wave_insts_per_launch_from_committed_profile": ins, "wave_insts_per_s": (ins / (per[iv] * 1e-3)) if ins and own_iv else None,
                         "row_loop_mix": km.get("by_kind"), "cycles_per_valu_inst_of_that_mix": cyc,
                         "issue_ceiling_wave_insts_per_s": ceil_k, "frac_of_issue_ceiling": (ins / (per[iv] * 1e-3) / ceil_k) if ins and own_iv else None,
                         **({} if own_iv else {"note": "runs beside the neighbouring classes on a side stream: no interval of its own"})})
        roof["valu"] = valu
        out = {
            "metric": "reads/sec remapped (%d bp PE)" % a.read_len, "value": 2.0 * a.pairs * world * a.steps / dt, "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 (int8 SW lanes, u64 hashes)", "data": "synthetic",
            "config": {"workload": "%s, %d x 2 x %d bp PE reads per GPU per step, preset sr" % (WORKLOADS.get(a.config, a.config), a.pairs, a.read_len), "key": a.config,
                       "reads_per_step_per_gpu": 2 * a.pairs, "pairs_asked_per_gpu": pairs_asked, "read_len": a.read_len, "sharding": "one input of %d fragments, rank r maps the contiguous range [r N / R, (r + 1) N / R); index replicated" % (a.pairs * world)},
            "value_incl_pcie": 2.0 * a.pairs / dt_io, "incl_pcie_note": "this rank: 4-bit packing on %d host threads + H2D, one step, D2H of %d records (%.1f MB); serial, no overlap between batches" % (min(32, os.cpu_count() or 1), int(nrec.value), nby.value / 1e6),
            "roofline": roof,
            "stages_ms": per,
            "merged_output": merged,
            "counters": {"minimizers_per_read": st.n_mini / (2.0 * a.pairs), "anchors_per_pair": st.n_anchor / float(a.pairs), "chains_per_pair": st.n_chain / float(a.pairs),
                         "regions_aligned_per_read": st.n_regs_aln / (2.0 * a.pairs), "ref_bases_per_region": st.n_refbases / max(1.0, float(st.n_regs_aln)),
                         "rechain": int(st.n_rechain), "heap_fallback": int(st.n_heap_fallback), "chain_fallback": int(st.n_chain_fallback), "side_stream_ms": float(st.ms_side_stream), "sort_tie_flags": int(st.n_sort_tie_flag), **(dist_info or {})},
            "host": {"gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "index_build_on_gpu_s": t_index, "pack_upload_s": t_upload, "reference_generation_s": t_gen, "read_simulation_s": t_reads, "reference_model": getattr(ref, "stats", None)},
        }
        if world == 1 and not a.no_cpu_baseline:
            n_cpu = max(1, a.cpu_sample_pairs)
            arr_cpu = arr if n_cpu <= a.pairs else np.concatenate([arr[:a.pairs], make_workload(a.config, a.pairs, n_cpu, a.read_len, 20261002, ref, a.ins_mean)])   # (the same input's next fragments: the sample stays the head of the file-to-file leg's files)
            out["cpu_baseline"], out["e2e_cli"] = cpu_baseline(tmp, "ref.fa", arr_cpu, n_cpu, a.read_len)
            out["parity_sample"] = {"pairs": n_cpu, "identical": out["e2e_cli"]["identical_sam"]}
            del arr_cpu
        if world == 1 and a.f2f_pairs > 0:
            out["file_to_file"] = file_to_file(tmp, "ref.fa", a, arr, ref, out["value"])
        print(json.dumps(out))
    shutil.rmtree(tmp, ignore_errors=True)
    if dist is not None:
        dist.barrier()
        if rank == 0 and shared:
            shutil.rmtree(shared, ignore_errors=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
