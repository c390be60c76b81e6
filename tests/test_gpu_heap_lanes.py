"""k_anchor_heap_lanes (the reference's binary-heap merge with the heap in the lanes of a wavefront, al_kernels_seed.hip) against a host
restatement of collect_seed_hits_heap (map.c:149-213, ksort.h:43-59) on random occurrence lists: up to 126 lists per fragment, many identical
lists (equal heads: the pop order among them is heap-shape dependent), lists longer than the prefetch ring.  The harness
(tests/csrc/heap_lanes_test.hip) includes the kernel source and is built here with hipcc."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hl") / "heap_lanes_test")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-two-entry-phi-node-folding-threshold=200", "-I", os.path.join(ROOT, "airlift_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "csrc", "heap_lanes_test.hip")], check=True)
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 8, 11])
def test_heap_in_lanes_equals_the_reference_heap(harness, seed):
    r = subprocess.run([harness, "2000", str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ": 0 differ" in r.stdout


@pytest.mark.gpu
def test_heap_in_lanes_long_lists(harness):
    """one fragment, 100 lists of 2500 positions (ring refills all the way, two register sets)"""
    r = subprocess.run([harness, "1", "3", "100", "2500"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
