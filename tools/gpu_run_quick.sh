#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
set -u
export TMPDIR=/tmp
O=gpurun_out/q
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/trace -- python tools/bench_family.py improved48 1 > $O/fam.txt 2>&1
grep -a "improved48 B" $O/fam.txt
python tools/rocprof_tail.py $O/trace 75 > $O/tail.txt
rm -rf $O/trace
