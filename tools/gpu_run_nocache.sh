#!/bin/bash
# Memory-safety pass: every GPU test file in its own process with PyTorch's caching allocator off (every tensor its own
# hipMalloc: an out-of-bounds access of a kernel lands outside its buffer instead of inside a cached block), and a guard
# pattern behind every workspace (an overrun that stays inside the allocation's page is caught too).
set -u
O=gpurun_out/${1:-nocache}
mkdir -p $O
export PYTORCH_NO_CUDA_MEMORY_CACHING=1
export FSN_WS_CANARY=1  # a guard pattern behind every workspace, verified after every test (fullsubnet_amd/_lib.py)
for f in tests/test_gpu_*.py; do
  n=$(basename $f .py)
  # (stream capture cannot allocate, and every hipFree synchronises the device: the captured-graph test and the test of
  # launches that share the chip from several streams need the caching allocator)
  timeout 1500 python -m pytest $f -m gpu -q -k "not hip_graph and not graphed and not several_streams and not graph_replays" > $O/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|Fatal|error' $O/$n.log | tail -1)"
done
