"""Per-chunk latency of the streaming enhancer on one MI355X: python tools/bench_streaming.py [batch] [frames_per_chunk]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fullsubnet_amd.streaming import StreamingEnhancer  # noqa: E402
from fsn_synthetic import make_noisy, make_params  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
model = fullsubnet_amd.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0,
                             sb_num_neighbors=15, fb_output_activate_function="ReLU",
                             sb_output_activate_function=False, fb_model_hidden_size=512, sb_model_hidden_size=384,
                             norm_type="cumulative_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
model.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()})
model = model.cuda().eval()
L = max(16000 * 4, 256 * K * 40)  # at least 40 calls
noisy = torch.from_numpy(make_noisy(B, L, seed=1)).cuda()
enh = StreamingEnhancer(model, batch_size=B)
chunk = 256 * K
times = []
for pos in range(0, L, chunk):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = enh.process(noisy[:, pos:pos + chunk])
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
enh.flush()
steady = sorted(times[10:]) or sorted(times)
med, p99 = steady[len(steady) // 2], steady[max(int(len(steady) * 0.99) - 1, 0)]
print(f"streaming B={B}, {K} frame(s) = {chunk / 16:.0f} ms of audio per call: median {med * 1e3:.3f} ms, "
      f"p99 {p99 * 1e3:.3f} ms per call -> {chunk / 16000 / med:.1f} x real time per stream, "
      f"algorithmic latency {(2 + 1 + K) * 16} ms")
