#!/bin/bash
# multi-set group kernel: test + config 5 at batches around 32
set -u
O=gpurun_out/${1:-r03n}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_family.py -m gpu -q -x -k "several_sequence or improved" 2>&1 | tail -15
for B in 32 28 24 16; do timeout 120 python tools/bench_family.py improved48 $B 2>&1 | tail -1; done
