#!/bin/bash
# Round 5: A/B of the bench line under two environments (no extras, no CPU leg).  usage: gpu_run_r05_ab.sh <tag> "<env A>" "<env B>"
set -u
O=gpurun_out/${1:-r05ab}
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then E="$2"; else E="$3"; fi
    env $E timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$v$rep.json 2> $O/bench_$v$rep.err
    python - <<PY
import json
d = json.loads(open("$O/bench_$v$rep.json").read().strip().splitlines()[-1])
print("$v$rep [$E]", d["ms_per_step"], d["roofline"]["frac"], {k: v for k, v in d.get("stages_ms", d.get("stage_ms", {})).items() if k.startswith("sb_rec")}, d.get("parity", {}).get("max_abs_err"))
PY
  done
done
