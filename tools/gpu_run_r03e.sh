#!/bin/bash
set -u
O=gpurun_out/${1:-r03e}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_amp.py tests/test_gpu_train.py -m gpu -q -rP) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=|margins|^E  |FAILED|vs the emulation|worst deviation| vs fsn_train" $O/pytest.log | tail -60
for A in f32 f16 bf16; do timeout 300 python tools/bench_train.py 16 $A 2>&1 | tail -1; done | tee $O/train_times.txt
B="python tools/bench_train.py 16 f16"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $B > $O/trace.log 2>&1
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_train_f16.md "rocprofv3 --kernel-trace --stats -- $B" && head -25 $O/kernel_stats_train_f16.md | cut -c1-160
rm -rf $O/trace
