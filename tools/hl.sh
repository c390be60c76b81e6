#!/bin/bash
# builds and runs the heap-in-lanes harness on the GPU box: correctness on random lists, then the cost of a pop
mkdir -p gpurun_out/hl
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -two-entry-phi-node-folding-threshold=200 -I airlift_amd/csrc -I include -o /tmp/hl_test tests/csrc/heap_lanes_test.hip 2>&1 | grep -E "error"
{ for s in 7 8; do timeout 120 /tmp/hl_test 2000 $s; done
  timeout 120 /tmp/hl_test 1 3 40 4000; timeout 120 /tmp/hl_test 1 3 60 4000; timeout 120 /tmp/hl_test 1 3 70 4000; timeout 120 /tmp/hl_test 1 3 120 2500; timeout 120 /tmp/hl_test 2048 3 40 100; timeout 120 /tmp/hl_test 2048 3 100 100; } 2>&1 | tee gpurun_out/hl/hl.log
