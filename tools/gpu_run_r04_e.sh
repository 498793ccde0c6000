#!/bin/bash
# Round 4: the fused training graph (train_glue_kernels.hip): parity, timing, kernel trace of the fp32 and AMP steps.
set -u
O=gpurun_out/${1:-r04i}
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_amp.py tests/test_gpu_trainer.py -m gpu -q -rP -x) > $O/pytest_train.log 2>&1
echo "pytest rc=$?" >> $O/pytest_train.log
grep -E "passed|failed|rc=|^E  |FAILED|gradient margins|worst deviation" $O/pytest_train.log | tail -14
for A in f32 f16; do
  C="python tools/bench_train.py 16 $A"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$A -- $C > $O/train_$A.txt 2>&1
  DB=$(ls $O/trace_$A/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_train_$A.md "rocprofv3 --kernel-trace --stats -- $C"
  grep "train step" $O/train_$A.txt
  rm -rf $O/trace_$A
  echo "at::native kernels in the $A trace:"; grep -c "at::native" $O/kernel_stats_train_$A.md
done
