#!/bin/bash
# Round 4: the BPTT group kernel under the fp32 arithmetic: parity (training tests), probe, step timing A/B.
set -u
O=gpurun_out/${1:-r04g}
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_amp.py -m gpu -q -rP -x) > $O/pytest_train.log 2>&1
echo "pytest rc=$?" >> $O/pytest_train.log
grep -E "passed|failed|rc=|^E  |FAILED|gradient margins|worst deviation" $O/pytest_train.log | tail -12
tools/bin/probe_g16_ar0 193 32 2 2>&1 | tee $O/probe_g16_f32.txt
timeout 300 python tools/bench_train.py 16 f32 2>&1 | tail -1
timeout 300 python tools/bench_train.py 16 f32 g16=0 2>&1 | tail -1
timeout 300 python tools/bench_train.py 16 f16 2>&1 | tail -1
