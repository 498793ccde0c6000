#!/bin/bash
set -u
O=gpurun_out/${1:-r03c}
mkdir -p $O
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q -rP) > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|rc=|margins|hog |^E  |FAILED|training step|DDP 2" $O/pytest.log | tail -40
(time timeout 600 python bench.py --steps 10 --warmup 3) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"], d["cpu_baseline"]["by_threads"])
PY
tail -4 $O/bench.err
