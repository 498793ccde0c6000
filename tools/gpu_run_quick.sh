#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
set -u
export TMPDIR=/tmp
O=gpurun_out/q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_amp.py tests/test_gpu_trainer.py tests/test_gpu_residency.py -m gpu -x -q -rP 2>&1 | grep -E "passed|failed|margins|^E " | tail -6
for A in f32 f16; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$A -- python tools/bench_train.py 16 $A > $O/train_$A.txt 2>&1
  DB=$(ls $O/trace_$A/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_train_$A.md "x"
  grep -a "train step" $O/train_$A.txt
  grep "fb_chain" $O/kernel_stats_train_$A.md | cut -c1-100
  rm -rf $O/trace_$A
done
