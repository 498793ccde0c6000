#!/bin/bash
# Config 5 at batch 1: kernel trace + wall time (where do the 6.9 ms go?)
set -u
O=gpurun_out/${1:-r03m}
mkdir -p $O
export TMPDIR=/tmp
for B in 1 4; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$B -- python tools/bench_family.py improved48 $B > $O/fam_$B.txt 2>&1
  DB=$(ls $O/trace_$B/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_improved48_b$B.md "rocprofv3 --kernel-trace --stats -- python tools/bench_family.py improved48 $B"
  tail -1 $O/fam_$B.txt
  rm -rf $O/trace_$B
done
python tools/bench_family.py improved48 1
python tools/bench_family.py improved48 2
