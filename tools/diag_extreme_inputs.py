import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import fullsubnet_amd as fsn
from fsn_synthetic import make_noisy, make_params
sys.path.insert(0, "/root/repo/oracle")
m = fsn.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15, fb_output_activate_function="ReLU",
              sb_output_activate_function=False, fb_model_hidden_size=512, sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
m.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()}); m = m.cuda().eval()
L = 16000
cases = {"silence": torch.zeros(2, L), "tiny": 1e-20 * torch.randn(2, L), "loud": 1e4 * torch.randn(2, L), "huge": 1e18 * torch.randn(2, L),
         "dc": torch.ones(2, L), "mixed": torch.cat([torch.zeros(1, L), torch.randn(1, L)])}
for name, x in cases.items():
    y = m.enhance(x.cuda())
    # reference algebra through the staged path (ATen ops on the same kernels) as a sanity: finite?
    print(f"{name:8s}: finite {bool(torch.isfinite(y).all())}, max |y| {float(y.abs().max()):.3e}, input max {float(x.abs().max()):.3e}")
