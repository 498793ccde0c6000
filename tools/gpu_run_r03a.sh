#!/bin/bash
# GPU session A of round 3: full GPU suite (with the residency tests), the bench line with the new side figures,
# kernel traces of the sibling models.  Everything lands under gpurun_out/$1/.
set -u
O=gpurun_out/${1:-r03a}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q -rP -x) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=|margins|hog " $O/pytest.log | tail -30
(time timeout 900 python bench.py --steps 10 --warmup 3) > $O/bench.json 2> $O/bench.err
tail -c 2500 $O/bench.json
tail -5 $O/bench.err
for W in "fast 256" "improved48 32"; do
  set -- $W
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$1 -- python tools/bench_family.py $1 $2 > $O/fam_$1.txt 2>&1
  DB=$(ls $O/trace_$1/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocprof_summary.py $DB $O/kernel_stats_$1_b$2.md "rocprofv3 --kernel-trace --stats -- python tools/bench_family.py $1 $2"
  tail -2 $O/fam_$1.txt
  rm -rf $O/trace_$1
done
