#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/gru6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gru_rows.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -16 > $O/tests.txt
tail -4 $O/tests.txt
timeout 300 python tools/bench_family.py gru 64 2>&1 | tail -1 | tee $O/bench_gru64.txt
