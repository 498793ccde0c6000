#!/bin/bash
# scratch session: edit, run, read (kept as the one ad-hoc runner)
export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_gpu_gru_rows.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tee -a $O/gru_soak.txt
done
python - <<'PY' 2>&1 | tee -a gpurun_out/quick/gru_soak.txt
import sys, time, torch
sys.path.insert(0, "tools")
import bench_family as BF
pack = BF.build("gru")
model = pack[0]
enh = BF.enhance_fn("gru", model)
from fsn_synthetic import make_noisy
y = torch.from_numpy(make_noisy(8, 48000, seed=1)).cuda().repeat(8, 1).contiguous()
ref = enh(y).clone()
ts = []
for i in range(300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = enh(y)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if not torch.equal(out, ref):
        print("MISMATCH at call", i); break
ts = sorted(ts)
print(f"GRU FullSubNet 64 x 3 s, 300 calls: bit-identical to the first = {torch.equal(out, ref)}; ms median {ts[150]:.2f}, min {ts[0]:.2f}, max {ts[-1]:.2f}")
PY
