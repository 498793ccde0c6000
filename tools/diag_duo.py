"""lstm2_duo_kernel (one workgroup per CU, both layers) against lstm2_group_kernel (two per CU): bit-identity of the mask and
the enhanced waveform, and the time of a call, at the batch sizes of the strong-scaling shares.  usage: diag_duo.py [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    fullsubnet_amd._lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from fullsubnet_amd import _lib  # noqa: E402
from fsn_synthetic import make_noisy, make_params  # noqa: E402

L = _lib.lib()
model = fullsubnet_amd.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                             fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                             sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
model.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=0, gain=2.0, mask_gain=24.0).items()})
model = model.cuda().eval()
length = int(os.environ.get("DUO_LEN", "48000"))
for B in [int(a) for a in sys.argv[1:]] or [8, 16, 6, 12]:
    x = torch.from_numpy(make_noisy(B, length, seed=5)).cuda()
    res = {}
    for on in (0, 1, 0, 1):
        L.fsn_debug_group_duo(on)
        for _ in range(2):
            enh, crm = model.enhance(x, return_crm=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            enh, crm = model.enhance(x, return_crm=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        res.setdefault(on, []).append((ms, enh.clone(), crm.clone()))
    plan = _lib.core_plan(model._cfg, B, 1 + length // 256)
    same = torch.equal(res[0][0][2], res[1][0][2]) and torch.equal(res[0][0][1], res[1][0][1])
    d = (res[0][0][2] - res[1][0][2]).abs().max().item()
    print(f"B={B}: group {res[0][0][0]:.2f} / {res[0][1][0]:.2f} ms, duo {res[1][0][0]:.2f} / {res[1][1][0]:.2f} ms, bit-identical {same} "
          f"(max |d mask| {d:.2e}, finite {bool(torch.isfinite(res[1][0][2]).all())}), clusters {plan['group_clusters']}", flush=True)
L.fsn_debug_group_duo(1)
