#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
timeout 600 python -m pytest tests/test_gpu_family.py -m gpu -x -q -k "fast or fullband or sequence_model" 2>&1 | tail -2
for B in 1 4 8 256; do timeout 120 python tools/bench_family.py fast $B 2>&1 | tail -1; done
