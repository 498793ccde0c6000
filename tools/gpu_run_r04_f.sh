#!/bin/bash
# Round 4: gemm_tn16 (16-bit weight-gradient products): parity (AMP tests) + kernel trace of the AMP step.
set -u
O=gpurun_out/${1:-r04j}
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_amp.py -m gpu -q -rP -x) > $O/pytest_amp.log 2>&1
echo "pytest rc=$?" >> $O/pytest_amp.log
grep -E "passed|failed|rc=|^E  |FAILED|worst deviation|margins" $O/pytest_amp.log | tail -14
C="python tools/bench_train.py 16 f16"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $C > $O/train_f16.txt 2>&1
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_train_f16.md "rocprofv3 --kernel-trace --stats -- $C"
grep "train step" $O/train_f16.txt
head -14 $O/kernel_stats_train_f16.md | cut -c1-140
rm -rf $O/trace
timeout 300 python tools/bench_train.py 16 f16 2>&1 | tail -1
