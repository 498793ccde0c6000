#!/bin/bash
# Round 5 probe session: GPU suite + the scheduling probes named on the command line (tools/bin/*, built here by hipcc).
set -u
O=gpurun_out/${1:-r05probe}
shift || true
mkdir -p $O
export TMPDIR=/tmp
if [ "${SKIP_PYTEST:-0}" != "1" ]; then
  (time timeout 1500 python -m pytest tests -m gpu -q -rP ${PYTEST_ARGS:-}) > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log
  grep -E "passed|failed|rc=|^E  |FAILED|margins" $O/pytest.log | tail -14
fi
for P in "$@"; do
  echo "== $P"
  timeout 300 tools/bin/$P > $O/$P.txt 2>&1
  cat $O/$P.txt
done
