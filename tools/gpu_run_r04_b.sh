#!/bin/bash
# Round 4, session B: the 16-bit arithmetic's own group kernels (lstm_group16_kernels.hip): parity (AMP tests), timing A/B,
# kernel trace of the AMP training step.
set -u
O=gpurun_out/${1:-r04b}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 600 python -m pytest tests/test_gpu_amp.py -m gpu -q -rP -x) > $O/pytest_amp.log 2>&1
echo "pytest rc=$?" >> $O/pytest_amp.log
grep -E "passed|failed|rc=|^E  |FAILED|vs the emulation|worst deviation" $O/pytest_amp.log | tail -40
timeout 300 python tools/bench_train.py 16 f16 2>&1 | tail -2
timeout 300 python tools/bench_train.py 16 f16 g16=0 2>&1 | tail -2
C="python tools/bench_train.py 16 f16"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_f16 -- $C > $O/train_f16.txt 2>&1
DB=$(ls $O/trace_f16/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_train_f16.md "rocprofv3 --kernel-trace --stats -- $C"
head -12 $O/kernel_stats_train_f16.md
rm -rf $O/trace_f16
