#!/bin/bash
# GPU suite (optional) + the bench line.  usage: gpu_run_bench.sh <tag> [pytest args...]
set -u
O=gpurun_out/${1:-bench}
shift || true
mkdir -p $O
export TMPDIR=/tmp
if [ "${SKIP_PYTEST:-0}" != "1" ]; then
  (time timeout 1500 python -m pytest tests -m gpu -q -rP "$@") > $O/pytest.log 2>&1
  echo "pytest rc=$?" >> $O/pytest.log
  grep -E "passed|failed|rc=|^E  |FAILED|margins" $O/pytest.log | tail -14
fi
(time timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:-}) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "roofline", "parity") if k in d})
for k in ("stages_ms", "train_step", "train_step_amp", "fast_b256", "improved48_b32", "cpu_baseline"):
    if k in d: print(k, d[k])
PY
tail -3 $O/bench.err
