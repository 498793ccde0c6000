#!/bin/bash
# Round 4: the sibling models (configs 4 / 5): family parity tests + kernel trace of one workload.
# usage: gpu_run_r04_h.sh <out> "<workload>" [pytest -k expr]
set -u
O=gpurun_out/${1:-r04s}
W=${2:-fast 256}
K=${3:-fast}
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_family.py -m gpu -q -x -k "$K") > $O/pytest_family.log 2>&1
echo "pytest rc=$?" >> $O/pytest_family.log
grep -E "passed|failed|rc=|^E  |FAILED" $O/pytest_family.log | tail -8
C="python tools/bench_family.py $W"
N=$(echo $W | tr ' ' '_')
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- $C > $O/bench_$N.txt 2>&1
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_$N.md "rocprofv3 --kernel-trace --stats -- $C"
tail -2 $O/bench_$N.txt | cut -c1-600
head -24 $O/kernel_stats_$N.md | cut -c1-150
rm -rf $O/trace
timeout 300 $C 2>&1 | tail -1 | cut -c1-400
