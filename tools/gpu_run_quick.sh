#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
(time timeout 1500 python -m pytest tests -x -q -m gpu) > $O/pytest_final.log 2>&1; echo "rc=$?" >> $O/pytest_final.log; grep -E "passed|failed|rc=" $O/pytest_final.log | tail -2
