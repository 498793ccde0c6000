#!/bin/bash
# Round 5: A/B of the bench line under several environments (no extras, no CPU leg).  usage: gpu_run_r05_ab.sh <tag> "<env A>" "<env B>" ...
set -u
O=gpurun_out/${1:-r05ab}
shift
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$i$rep.json 2> $O/bench_$i$rep.err
    python - <<PY
import json
d = json.loads(open("$O/bench_$i$rep.json").read().strip().splitlines()[-1])
print("[$E]", d["ms_per_step"], d["roofline"]["frac"], {k: v for k, v in d["stage_ms"].items() if k.startswith("sb_")}, d["parity"]["max_abs_err_cirm_vs_oracle"])
PY
  done
done
