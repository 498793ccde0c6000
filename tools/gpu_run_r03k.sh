#!/bin/bash
set -u
O=gpurun_out/${1:-r03k}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_family.py -m gpu -q -rP -k "improved") > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^E  |FAILED" $O/pytest.log | tail -10
for W in "improved48 32" "improved48 8" "improved48 4" "improved48 1" "improved769 32" "improved16 32"; do set -- $W; timeout 200 python tools/bench_family.py $1 $2 2>&1 | tail -1; done | tee $O/fam.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python tools/bench_family.py improved48 32 > $O/fam_prof.txt 2>&1
DB=$(ls $O/trace/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_improved48_b32.md "rocprofv3 --kernel-trace --stats -- python tools/bench_family.py improved48 32" && head -12 $O/kernel_stats_improved48_b32.md | cut -c1-150
rm -rf $O/trace
