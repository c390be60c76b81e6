"""-m gpu: bench.py end to end on a small workload -- the single-GPU line carries what the contract asks for, and the N > 1 path
(one input sharded by rank, offsets exchanged once per step, max-over-ranks timing) runs with two ranks on this one GPU
(--test-one-gpu: device 0 for both, gloo collectives; on a multi-GPU node the same code runs one rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    for l in reversed(out.decode().split("\n")):
        if l.startswith("{"):
            return json.loads(l)
    raise AssertionError("no JSON line in: " + out.decode()[-2000:])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--pairs", "200000", "--steps", "2", "--warmup", "1", "--cpu-sample-pairs", "50000"], capture_output=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["key"] == "c2" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and len(rf["stages"]) >= 6
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("reference", "port") and cb["cores"] >= 1
    assert d["parity_sample"]["identical"] is True


def test_bench_two_ranks_shard_one_input():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--test-one-gpu", "--config", "c2", "--pairs", "100000", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    m = d["merged_output"]
    assert m["rec_off"] == 0 and m["byte_off"] == 0 and m["records"] > 2 * 100000 and m["bytes"] > 48 * m["records"] - 1      # rank 0's block starts the merged output; totals cover both ranks
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--pairs", "200000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, cwd=ROOT)
    d1 = _line(one.stdout)
    assert d1["merged_output"]["records"] == m["records"]            # two shards of the one input == the whole input
