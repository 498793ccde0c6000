#!/bin/bash
# Run the commands given as arguments one after the other, logs under gpurun_out/<tag>/ (first argument).
set -u
O=gpurun_out/${1:-cmds}
shift || true
mkdir -p $O
export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  echo "== [$i] $C"
  (time timeout 1200 bash -c "$C") > $O/cmd$i.log 2>&1
  echo "rc=$?" >> $O/cmd$i.log
  grep -E "passed|failed|rc=|^E  |FAILED|train step|ms per|worst deviation|margins" $O/cmd$i.log | tail -${TAILN:-12}
done
