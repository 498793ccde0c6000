"""Inference time against the utterance length (one and eight utterances): python tools/bench_length_sweep.py [seconds ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fsn_synthetic import make_noisy, make_params  # noqa: E402

m = fullsubnet_amd.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                         fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                         sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()})
m = m.cuda().eval()
for sec in [float(a) for a in sys.argv[1:]] or [1, 3, 10, 30, 60, 90]:
    L = int(16000 * sec)
    for B in (1, 8):
        x = torch.from_numpy(make_noisy(B, L, seed=1)).cuda()
        for _ in range(2):
            y = m.enhance(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = m.enhance(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{sec:5.1f} s x {B}: {dt * 1e3:8.2f} ms = {B * sec / dt:7.0f} x real time, {dt * 1e6 / (1 + L // 256):.1f} us per frame, finite {bool(torch.isfinite(y).all())}")
