"""Which ATen operators still launch device kernels inside a model call (round 5: Improved / Fast FullSubNet glue)?
python tools/diag_aten_ops.py improved48 32   - torch.profiler over one call, device-kernel-launching aten ops with the
source line that issued them."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_family as BF  # noqa: E402
from fsn_synthetic import make_noisy  # noqa: E402

which, batch = sys.argv[1], int(sys.argv[2])
pack = BF.build(which, torch.device("cuda"))
model, _, L, hop, sr, la = pack
fn = BF.enhance_fn(which, model)
x = torch.from_numpy(make_noisy(batch, L, seed=1)).cuda()
for _ in range(2):
    fn(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fn(x)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    dev = getattr(ev, "self_device_time_total", 0) or 0
    if not ev.name.startswith("aten::") or dev <= 0:
        continue
    where = "?"
    for st in (ev.stack or []):
        if "fullsubnet_amd/" in st or "bench_family" in st:
            where = st.split("/")[-1][:80]
            break
    r = rows.setdefault((ev.name, where), [0, 0.0])
    r[0] += 1
    r[1] += dev
for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} x{n:3d} {us:8.1f} us  {where}")
print(f"{sum(r[0] for r in rows.values())} device-kernel-launching aten calls, {sum(r[1] for r in rows.values()):.0f} us")
