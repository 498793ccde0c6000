#!/bin/bash
# Round 4, session A: the GPU suite (new: config-4 full size, 769-bin golden, two-rank bench, persistent kernels in a
# two-rank job) and the bench line with its self-checking side figures.
set -u
O=gpurun_out/${1:-r04a}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q -rP) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|^E  |FAILED|^fast B=" $O/pytest.log | tail -14
(time timeout 900 python bench.py --steps 10 --warmup 3) > $O/bench.json 2> $O/bench.err
tail -c 2500 $O/bench.json; tail -3 $O/bench.err
