#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
timeout 900 python -m pytest tests/test_gpu_family.py -m gpu -x -q -rP -k "improved" 2>&1 | grep -E "passed|failed|section input|config 5|^E " | tail -8
for B in 1 2 4 16 32; do timeout 120 python tools/bench_family.py improved48 $B 2>&1 | tail -1; done
timeout 120 python tools/bench_family.py improved16 1 2>&1 | tail -1
