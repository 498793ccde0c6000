#!/bin/bash
# Round 6 session A: probe table (with the dwordx2 controls), GPU suite twice, the bench line.
set -u
O=gpurun_out/${1:-r06a}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 600 tools/bin/probe_store_hazard $O/store_hazard.csv) > $O/store_hazard.md 2> $O/store_hazard.err
echo "probe rc=$?"; tail -4 $O/store_hazard.md
for rep in 1 2; do
  (time timeout 1800 python -m pytest tests -m gpu -q -rP) > $O/pytest_$rep.log 2>&1
  echo "pytest rc=$?" >> $O/pytest_$rep.log
  grep -E "passed|failed|rc=|^E  |FAILED" $O/pytest_$rep.log | tail -8
done
(time timeout 900 python bench.py --steps 10 --warmup 3) > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; tail -3 $O/bench.err
