#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/gru8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gru_rows.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -30 | tee $O/tests.txt
for B in 16 17 18; do
  timeout 300 python tools/bench_family.py gru $B 2>&1 | tail -1 | tee -a $O/bench_gru_sweep.txt
done
