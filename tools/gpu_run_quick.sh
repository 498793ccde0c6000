#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_gpu_residency.py -m gpu -x -q 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
