#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
(time timeout 1500 python -m pytest tests -m gpu -q -x) 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=%d: %.2f ms per batch, %.0f frames/s frac %.4f' % (d['config']['batch_total'], d['ms_per_step'], d['value'], d['roofline']['frac']))"
