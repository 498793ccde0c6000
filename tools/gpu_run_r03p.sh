#!/bin/bash
# Config 5 after the multi-set group launch: kernel traces at batch 32 and 1
set -u
O=gpurun_out/${1:-r03p}
mkdir -p $O
export TMPDIR=/tmp
for B in 32 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$B -- python tools/bench_family.py improved48 $B > $O/fam_$B.txt 2>&1
  DB=$(ls $O/trace_$B/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_improved48_b$B.md "rocprofv3 --kernel-trace --stats -- python tools/bench_family.py improved48 $B"
  tail -1 $O/fam_$B.txt
  rm -rf $O/trace_$B
done
