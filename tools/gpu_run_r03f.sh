#!/bin/bash
set -u
O=gpurun_out/${1:-r03f}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_amp.py -m gpu -q -rP) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=|margins|^E  |FAILED|vs the emulation|worst deviation| vs fsn_train" $O/pytest.log | tail -60
for A in f16 bf16; do timeout 300 python tools/bench_train.py 16 $A 2>&1 | tail -1; done | tee $O/train_times.txt
