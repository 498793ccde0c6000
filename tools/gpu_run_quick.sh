#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for B in 8 10 12 16 20 24 32 40 48 56 64 72; do
  python bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d: %.2f ms per batch, %.0f frames/s' % (d['config']['batch_total'], d['ms_per_step'], d['value']))"
done
