#!/bin/bash
set -u
O=gpurun_out/${1:-r03b}
mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/diag_hog.py > $O/diag_hog.txt 2>&1
cat $O/diag_hog.txt | tail -20
(time timeout 900 python -m pytest tests/test_gpu_residency.py tests/test_gpu_train.py tests/test_gpu_trainer.py tests/test_gpu_streaming.py -m gpu -q -rP) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=|margins|hog |^E  |FAILED" $O/pytest.log | tail -40
