"""Hunt for intermittent stalls: repeat a small enhancement call (eager and as a hipGraph replay) and report calls that take far
longer than the median.  usage: diag_stall.py [which=improved16] [batch=3] [reps=400] [length=8192]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd as fsn  # noqa: E402
import bench_family  # noqa: E402
from fsn_synthetic import make_noisy  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "improved16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
L = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
if which == "fullsubnet":
    from fsn_synthetic import make_params
    m = fsn.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                  fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                  sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()})
    m = m.cuda().eval()
    fn = m.enhance
else:
    m = bench_family.build(which)[0]
    fn = bench_family.enhance_fn(which, m)
x = torch.from_numpy(make_noisy(B, L, seed=33)).cuda()
for mode in ("eager", "graph"):
    call = fn if mode == "eager" else fsn.GraphedCall(fn)
    for _ in range(3):
        y = call(x)
    torch.cuda.synchronize()
    ts, bad = [], 0
    for i in range(reps):
        t0 = time.perf_counter()
        y = call(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ts.append(dt)
        if not bool(torch.isfinite(y).all()):
            bad += 1
            print(f"  {mode} call {i}: NON-FINITE output after {dt * 1e3:.1f} ms")
    s = sorted(ts)
    med = s[len(s) // 2]
    slow = [(i, t) for i, t in enumerate(ts) if t > 20 * med]
    print(f"{which} B={B} {mode}: median {med * 1e3:.2f} ms, max {s[-1] * 1e3:.1f} ms, {len(slow)} of {reps} calls > 20 x median "
          f"{[(i, round(t * 1e3)) for i, t in slow[:6]]}, {bad} non-finite")
