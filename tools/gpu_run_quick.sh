#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
for B in 64 72; do
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace$B -- python tools/bench_family.py fast $B > $O/fam_fast_$B.txt 2>&1
DB=$(ls $O/trace$B/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_fast_b$B.md "rocprofv3 --kernel-trace --stats -- python tools/bench_family.py fast $B"
rm -rf $O/trace$B
done
