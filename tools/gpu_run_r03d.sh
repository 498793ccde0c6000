#!/bin/bash
set -u
O=gpurun_out/${1:-r03d}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q -rP) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=|margins|hog |^E  |FAILED|training step|DDP 2" $O/pytest.log | tail -40
timeout 300 python tools/bench_family.py fast 256 > $O/fam_fast.txt 2>&1; tail -1 $O/fam_fast.txt
timeout 300 python tools/bench_family.py fast 64 >> $O/fam_fast.txt 2>&1; tail -1 $O/fam_fast.txt
