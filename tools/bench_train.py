"""Time one training step (BASELINE config 3 per-rank shape: 16 x 3.072 s) on one MI355X.
python tools/bench_train.py [batch] [f32|f16|bf16] [g16=0] [overlap=0] [norm=cumulative] [saves=32]   (f16 / bf16: autocast arithmetic +
GradScaler; norm=cumulative: the shipped train_cumulativeLaplaceNorm.toml's norm)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fullsubnet_amd.train import train_step  # noqa: E402
from fsn_synthetic import make_noisy, make_params  # noqa: E402

B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 49152
ARITH = sys.argv[2] if len(sys.argv) > 2 else "f32"
model = fullsubnet_amd.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0,
                             sb_num_neighbors=15, fb_output_activate_function="ReLU",
                             sb_output_activate_function=False, fb_model_hidden_size=512, sb_model_hidden_size=384,
                             norm_type="cumulative_laplace_norm" if "norm=cumulative" in sys.argv else "offline_laplace_norm",
                             num_groups_in_drop_band=2, weight_init=False)
model.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()})
model = model.cuda().train()
model.train_arithmetic = ARITH
for a in sys.argv:
    if a.startswith("saves="):  # saves=32: the gates BPTT re-reads kept in fp32 (default under f16 / bf16: "16", FSN_ARITH_SAVES16)
        model.train_saves = a[6:]
if "g16=0" in sys.argv:  # A/B: the fp32-era group kernels under the 16-bit arithmetic (lstm_group16_kernels.hip off)
    fullsubnet_amd._lib.lib().fsn_debug_g16_kernels(0)
if "g16=2" in sys.argv:  # A/B: weight-gradient products converting their operands on the fly (gemm_tn16_kernel)
    fullsubnet_amd._lib.lib().fsn_debug_g16_kernels(2)
if "g16=3" in sys.argv:  # A/B: dx / dW_ih0 from fp32 gate gradients (the BPTT launch stores them), round 5's form
    fullsubnet_amd._lib.lib().fsn_debug_g16_kernels(3)
if "overlap=0" in sys.argv:  # A/B: the sub-band weight-gradient products in line instead of beside the full-band backward
    import fullsubnet_amd.train as _tr
    _tr.OVERLAP_WEIGHT_PRODUCTS = False
if "wide=0" in sys.argv:  # A/B: the 16-bit-operand products on 192 x 192 tiles instead of 192 x 384
    fullsubnet_amd._lib.lib().fsn_debug_tn16h_wide(0)
scaler = torch.amp.GradScaler("cuda", enabled=ARITH != "f32")
opt = fullsubnet_amd.ClipAdam(model.parameters(), lr=1e-3)
noisy = torch.from_numpy(make_noisy(B, L, seed=1)).cuda()
clean = torch.from_numpy(0.7 * make_noisy(B, L, seed=2)).cuda()
for _ in range(2):
    loss = train_step(model, opt, noisy, clean, scaler=scaler)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    loss = train_step(model, opt, noisy, clean, scaler=scaler)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
T = 1 + L // 256
flops = 3 * 2 * (3803648 + 128 * 1819392) * B * (T + 2)  # ~3x forward, F -> 128 sub-band bins
print(f"train step B={B} {ARITH}: {dt * 1e3:.1f} ms, loss {loss.item():.5f}, ~{flops / dt / 1e12:.1f} TFLOP/s "
      f"({B * T / dt:.0f} frames/s), peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
