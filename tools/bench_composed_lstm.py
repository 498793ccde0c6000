import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import fullsubnet_amd
from fsn_synthetic import make_noisy
from fullsubnet_amd.acoustics.feature import stft
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(3)
kw = dict(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
          fb_output_activate_function="ReLU", sb_output_activate_function="Tanh", fb_model_hidden_size=512,
          sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=True)
m = fullsubnet_amd.Model(**kw).cuda().eval()
assert not m._fused
y = torch.from_numpy(make_noisy(8, 48000, seed=1)).cuda().repeat((B + 7) // 8, 1)[:B].contiguous()
with torch.no_grad():
    mag = stft(y, 512, 256, 512, return_phase=False)[0].unsqueeze(1)
    for _ in range(2): out = m(mag)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = m(mag)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"composed LSTM FullSubNet (sb Tanh) B={B}: {dt*1e3:.2f} ms per model call, finite={bool(torch.isfinite(out).all())}")
