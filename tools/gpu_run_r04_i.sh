#!/bin/bash
# Round 4: training steps A/B (the sub-band weight-gradient products beside the full-band backward) + training parity tests.
set -u
O=gpurun_out/${1:-r04u}
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_amp.py -m gpu -q -x) > $O/pytest_train.log 2>&1
echo "pytest rc=$?" >> $O/pytest_train.log
grep -E "passed|failed|rc=|^E  |FAILED" $O/pytest_train.log | tail -8
for A in f16 f32; do
  timeout 300 python tools/bench_train.py 16 $A 2>&1 | tail -1
  timeout 300 python tools/bench_train.py 16 $A overlap=0 2>&1 | tail -1
done
