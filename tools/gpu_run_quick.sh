#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/gru7
mkdir -p $O
for B in 12 15 16 20 24 31 32 48; do
  timeout 300 python tools/bench_family.py gru $B 2>&1 | tail -1 | tee -a $O/bench_gru_sweep.txt
done
bash tools/gpu_run_nocache.sh gru7/nocache 2>&1 | tee $O/nocache.txt
