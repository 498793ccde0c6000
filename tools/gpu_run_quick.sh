#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
timeout 600 python -m pytest tests/test_gpu_family.py -m gpu -x -q -k "fast" 2>&1 | tail -2
for B in 256 320 384 512; do timeout 120 python tools/bench_family.py fast $B 2>&1 | tail -1; done
python - <<'PY'
import torch, sys
sys.path.insert(0, 'tools')
import bench_family as bf
from fsn_synthetic import make_noisy
pack = bf.build('fast'); model = pack[0]; enh = bf.enhance_fn('fast', model)
x = torch.from_numpy(make_noisy(8, 8000, seed=3)).cuda().repeat(50, 1)[:390].contiguous()
whole = enh(x); parts = torch.cat([enh(x[i:i+130]) for i in range(0, 390, 130)])
print('fast 390 utterances chunked vs 3 x 130: max |d| %.3g of %.3g' % ((whole - parts).abs().max().item(), parts.abs().max().item()))
PY
