#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
mkdir -p gpurun_out/gru1
timeout 900 python -m pytest tests/test_gpu_gru_rows.py -m gpu -q -x -s 2>&1 | tail -30 > gpurun_out/gru1/tests.txt
tail -30 gpurun_out/gru1/tests.txt
timeout 300 python tools/bench_family.py gru 64 2>&1 | tail -5 | tee gpurun_out/gru1/bench_gru64.txt
timeout 300 python tools/bench_family.py gru 32 2>&1 | tail -3 | tee gpurun_out/gru1/bench_gru32.txt
