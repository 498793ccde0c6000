#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_family.py -m gpu -q -x -s -k "left_over_tiles_on_the_persistent" 2>&1 | grep -v "^$" | tail -8 | tee $O/tests_lstm_left2.txt
