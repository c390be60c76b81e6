#!/usr/bin/env python3
"""bench.py -- reads/s remapped by the MI355X re-alignment hot path (BASELINE.json metric).

A "step" is one pass of the whole hot path (minimizer sketch -> seed lookup -> anchor sort -> chaining ->
banded SW extension -> MAPQ/pairing -> alignment records) over one batch of synthetic 150 bp paired-end reads that is
already resident in HBM.  Workload at N=1: BASELINE.json configs[1] (yeast-sized reference pair, 150 bp PE; the
synthetic stand-in C2 of SURVEY.md 8(d), generated here with fixed seeds).  N>1: one process per GPU, reads sharded by
rank (every rank holds the full index; weak scaling), one 16-byte RCCL all-gather per step for the merged-output offsets.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
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
             "c3": "C3: ce11-sized synthetic reference (6 contigs, 100.3 Mbp, 3000 planted repeats)",
             "c4": "C4: human-sized synthetic reference (24 contigs, 3.1 Gbp, 20000 planted duplications, 2 % N)"}
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured streaming ceiling


def make_workload(pairs, read_len, seed, ref, ins_mean=None):
    """ASCII reads for `pairs` fragments, concatenated fragment-major (mate1, mate2, mate1, ...)."""
    import gen_synth as g
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    out = np.empty((pairs, 2, read_len), dtype=np.uint8)
    chunk = 250_000
    for s in range(0, pairs, chunk):
        n = min(chunk, pairs - s)
        kw = dict(ins_mean=550, ins_sd=60, ins_hi=1000) if read_len >= 200 else {}      # 250 bp PE, insert N(550,60)
        if ins_mean:
            kw = dict(ins_mean=ins_mean, ins_sd=ins_mean // 10, ins_hi=1000)
        r1, r2 = g.simulate_pairs(ref, n, read_len, seed=seed + s, **kw)
        out[s:s + n, 0] = lut[r1]; out[s:s + n, 1] = lut[r2]
    return out


def write_fastq_sample(path, arr, mate, prefix="realigned_"):
    n, _, L = arr.shape
    q = b"I" * L
    with open(path, "wb") as f:
        f.write(b"".join(b"@" + (prefix + str(i)).encode() + b"\n" + arr[i, mate].tobytes() + b"\n+\n" + q + b"\n" for i in range(n)))


def cpu_baseline(tmp, ref_fa, arr, n_pairs):
    """Times the CPU comparator on a bounded sample of the same workload (rank 0, N=1 only), then the drop-in CLI on the
    same files: end-to-end wall clock (FASTA + FASTQ in, SAM out) and byte identity of the two SAM streams."""
    import hashlib
    cores = os.cpu_count() or 1
    write_fastq_sample(os.path.join(tmp, "cb_1.fq"), arr[:n_pairs], 0)
    write_fastq_sample(os.path.join(tmp, "cb_2.fq"), arr[:n_pairs], 1)
    write_fastq_sample(os.path.join(tmp, "one_1.fq"), arr[:1], 0)
    write_fastq_sample(os.path.join(tmp, "one_2.fq"), arr[:1], 1)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mm2ref")
    if os.path.exists(ref_bin):
        kind, cmd = "reference", [ref_bin, "-t", str(cores)]
    else:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "al_oracle"], check=True)
        kind, cmd = "port", [os.path.join(ROOT, "oracle", "al_oracle"), "-t", str(cores)]
    dn = open(os.devnull, "wb")

    def timed(c, out):
        t0 = time.time()
        with open(out, "wb") as f:
            subprocess.run(c, cwd=tmp, stdout=f, stderr=dn, check=True)
        return time.time() - t0

    def md5(path):
        h = hashlib.md5()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    t_idx = timed(cmd + [ref_fa, "one_1.fq", "one_2.fq"], os.path.join(tmp, "one.sam"))
    t_all = timed(cmd + [ref_fa, "cb_1.fq", "cb_2.fq"], os.path.join(tmp, "cpu.sam"))
    t_map = max(t_all - t_idx, 1e-6)
    base = {"value": 2 * n_pairs / t_map, "unit": "reads/s", "cores": cores, "kind": kind,
            "sample": "%d pairs x 150 bp of the same workload, SAM to a file, index build (%.2f s) subtracted, wall %.2f s" % (n_pairs, t_idx, t_all)}
    cli = os.path.join(ROOT, "airlift_amd", "bin", "airlift-align")
    nt = min(cores, 32)
    t_cli = timed([cli, "-ax", "sr", "-t", str(nt), ref_fa, "cb_1.fq", "cb_2.fq"], os.path.join(tmp, "gpu.sam"))
    e2e = {"wall_s": t_cli, "reads_per_s": 2 * n_pairs / t_cli, "host_threads": nt, "cpu_wall_s": t_all, "speedup_vs_cpu_wall": t_all / t_cli,
           "identical_sam": md5(os.path.join(tmp, "gpu.sam")) == md5(os.path.join(tmp, "cpu.sam")),
           "note": "whole process, cold start: FASTA parse + index build on the GPU + FASTQ parse + mapping + SAM text; CPU wall likewise includes its index build"}
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
    ap.add_argument("--pairs", type=int, default=2_000_000, help="fragments per GPU per step (C2: 2 M pairs)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample-pairs", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ins-mean", type=int, default=0, help="mean insert size override (short inserts make the mates overlap: equal-key anchors)")
    ap.add_argument("--config", default="c2", help="synthetic reference of tools/gen_synth.py: c2 (BASELINE configs[1], default), c2r (c2 + high-copy element family), c3 (100 Mbp), c4 (3.1 Gbp)")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    import gen_synth as g
    import airlift_amd as A
    L = A.load()
    L.al_batch_upload_flat.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.c_int64]
    L.al_batch_upload_flat.restype = C.c_int

    # reference pair stand-in (C2 = 16 contigs, 12.16 Mbp, 150 planted duplications); identical on every rank.
    # The index is built on this rank's GPU from the FASTA (al_idx_build_device), as the CLI does.
    rk, _ = g.CONFIGS[a.config]
    ref = g.make_reference(**rk)
    tmp = tempfile.mkdtemp(prefix="al_bench_")
    g.write_fasta(os.path.join(tmp, "ref.fa"), ref)
    t0 = time.time()
    idx = A.Index(fasta=os.path.join(tmp, "ref.fa"), on_device=local if world > 1 else 0)
    t_index = time.time() - t0
    arr = make_workload(a.pairs, a.read_len, 20261002 + 7919 * rank, ref, a.ins_mean)
    ctx = A.Context(idx, device=local if world > 1 else 0)
    L.al_ctx_set_threads(ctx.h, min(32, os.cpu_count() or 1))       # host packing threads (outside the timed region)
    nf = a.pairs
    n_segs = (C.c_int * nf)(*([2] * nf)); qlens = (C.c_int * (2 * nf))(*([a.read_len] * (2 * nf)))
    t0 = time.time()
    rc = L.al_batch_upload_flat(ctx.h, nf, n_segs, qlens, arr.ctypes.data_as(C.c_char_p), b"realigned_", nf * rank)
    if rc != 0:
        raise SystemExit("upload failed")
    t_upload = time.time() - t0
    ctx.n_frag, ctx.n_reads = nf, 2 * nf

    def step():
        ctx.run()
        if dist is not None:   # merged-output offsets: {n_records, n_bytes} per rank over RCCL/xGMI
            st = ctx.stat()
            mine = torch.tensor([int(st.n_regs_aln), int(st.bytes_out)], dtype=torch.int64, device=dev)
            allr = torch.empty(2 * world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allr, mine)

    for _ in range(a.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    stage_ms = np.zeros(24); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        st = ctx.stat()
        stage_ms += np.array(list(st.ms_kernel))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = ctx.stat()
    if rank == 0:
        names = [L.al_stage_name(i).decode() for i in range(st.n_stage)]
        per = {names[i]: float(stage_ms[i] / a.steps) for i in range(st.n_stage)}
        kern = {names[i]: L.al_stage_kernel(i).decode() for i in range(st.n_stage)}
        dom = max((k for k in per if kern[k]), key=lambda k: per[k])          # intervals that are exactly one kernel
        traffic, traffic_src = load_traffic(kern[dom])
        alg = float(st.algorithmic_bytes)
        achieved = alg / (per[dom] * 1e-3) / 1e9
        out = {
            "metric": "reads/sec remapped (150 bp PE)", "value": 2.0 * a.pairs * world * a.steps / dt, "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 (int8 SW lanes, u64 hashes)", "data": "synthetic",
            "config": {"workload": "%s, %d x 2 x %d bp PE reads per GPU per step, preset sr" % (WORKLOADS.get(a.config, a.config), a.pairs, a.read_len),
                       "reads_per_step_per_gpu": 2 * a.pairs, "read_len": a.read_len, "sharding": "reads sharded by rank, index replicated"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kern[dom], "interval": dom, "kernel_ms": per[dom], "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_read": alg / (2.0 * a.pairs),
                         "pipeline_GBps": alg / (sum(per.values()) * 1e-3) / 1e9},
            "stages_ms": per,
            "counters": {"minimizers_per_read": st.n_mini / (2.0 * a.pairs), "anchors_per_pair": st.n_anchor / float(a.pairs), "chains_per_pair": st.n_chain / float(a.pairs),
                         "regions_aligned_per_read": st.n_regs_aln / (2.0 * a.pairs), "ref_bases_per_region": st.n_refbases / max(1.0, float(st.n_regs_aln)),
                         "rechain": int(st.n_rechain), "heap_fallback": int(st.n_heap_fallback), "sort_tie_flags": int(st.n_sort_tie_flag)},
            "host": {"index_build_on_gpu_s": t_index, "pack_upload_s": t_upload},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], out["e2e_cli"] = cpu_baseline(tmp, "ref.fa", arr, min(a.cpu_sample_pairs, a.pairs))
            out["parity_sample"] = {"pairs": min(a.cpu_sample_pairs, a.pairs), "identical": out["e2e_cli"]["identical_sam"]}
        print(json.dumps(out))
    shutil.rmtree(tmp, ignore_errors=True)
    ctx.close(); idx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
