#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/r06s18
mkdir -p $O
# the driver's own sequence on the final tree: smoke, the GPU suite, the default bench line; then the 10-step bench line
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/pytest1.log 2>&1; echo "rc=$?" >> $O/pytest1.log; grep -E "passed|failed|rc=" $O/pytest1.log | tail -2
(time timeout 900 python bench.py) > $O/bench_default.json 2> $O/bench_default.err; grep real $O/bench_default.err
(time timeout 900 python bench.py --steps 10 --warmup 3) > $O/bench.json 2> $O/bench.err; grep real $O/bench.err
