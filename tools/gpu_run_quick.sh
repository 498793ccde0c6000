#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gru_rows.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -6 | tee $O/tests.txt
