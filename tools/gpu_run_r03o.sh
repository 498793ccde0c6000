#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_gpu_family.py -m gpu -q -x -k "norm or improved" 2>&1 | tail -3
for B in 1 1 2 4 16 32; do timeout 120 python tools/bench_family.py improved48 $B 2>&1 | tail -1; done
timeout 120 python tools/bench_family.py improved16 1 2>&1 | tail -1
timeout 120 python tools/bench_family.py fast 1 2>&1 | tail -1
timeout 120 python tools/bench_family.py fullband 1 2>&1 | tail -1
