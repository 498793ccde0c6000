import sys, time, torch
sys.path.insert(0, "/root/repo")
import fullsubnet_amd
from fsn_synthetic import make_noisy, make_params
m = fullsubnet_amd.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15, fb_output_activate_function="ReLU",
                         sb_output_activate_function=False, fb_model_hidden_size=512, sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()}); m = m.cuda().eval()
for B in [int(a) for a in sys.argv[1:]]:
    x = torch.from_numpy(make_noisy(min(B, 8), 48000, seed=1)).cuda().repeat((B + 7) // 8, 1)[:B].contiguous()
    for _ in range(2): m.enhance(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m.enhance(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"B={B}: {dt*1e3:.1f} ms per batch = {dt*1e3/B:.2f} ms per utterance")
