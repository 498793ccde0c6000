#!/bin/bash
# Round 4, probe session: parity of the 16-bit group kernels (AMP tests), ablation probe, step timing A/B.
set -u
O=gpurun_out/${1:-r04d}
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_amp.py -m gpu -q -rP -x) > $O/pytest_amp.log 2>&1
echo "pytest rc=$?" >> $O/pytest_amp.log
grep -E "passed|failed|rc=|^E  |FAILED|worst deviation" $O/pytest_amp.log | tail -12
tools/bin/probe_g16 195 32 ${2:-3} 2>&1 | tee $O/probe_g16.txt
timeout 300 python tools/bench_train.py 16 f16 2>&1 | tail -1
