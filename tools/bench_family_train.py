"""Time one training step of the sibling recipes that ship training TOMLs, on one MI355X:
python tools/bench_family_train.py [fast|fullband] [batch] [f32|f16|bf16]   (fast_fullsubnet/train_shrinkSize2.toml: batch 72 x
3.072 s, use_amp = true; fullband_baseline/train.toml).  The LSTM / Linear blocks run on the library's training entries, the glue
between them is autograd-tracked tensor algebra.  f16 / bf16: the trainer's autocast arithmetic + GradScaler -
Fast FullSubNet's bottleneck (two layers x 384 units on B x 64 rows) then runs on the 16-bit persistent training kernels."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fullsubnet_amd.train import train_step  # noqa: E402
from fsn_synthetic import make_fast_params, make_fullband_params, make_noisy  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fast"
B, L = int(sys.argv[2]) if len(sys.argv) > 2 else 72, 49152
ARITH = sys.argv[3] if len(sys.argv) > 3 else "f32"
if which == "fast":
    from fullsubnet_amd.fast_fullsubnet import Model
    model = Model(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
                  bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
                  encoder_output_num_neighbors=0, norm_type="offline_laplace_norm", weight_init=False)
    sd = {k: torch.from_numpy(v) for k, v in make_fast_params(seed=3).items()}
    sd["mel_scale.fb"] = model.mel_scale.fb.clone()
    # MAC per utterance and frame: encoder + decoder at the frame rate, bottleneck at half rate
    mac = 4 * 384 * (64 + 384) + 4 * 257 * (384 + 257) + 257 * 64 + 4 * 512 * (128 + 512) + 4 * 512 * 1024 + 512 * 514 \
        + 64 * (4 * 384 * (12 + 384) + 4 * 384 * 768 + 384) // 2
else:
    from fullsubnet_amd.fullband_baseline import Model
    model = Model(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=False, look_ahead=2,
                  norm_type="offline_laplace_norm", weight_init=False)
    sd = {k: torch.from_numpy(v) for k, v in make_fullband_params(seed=3).items()}
    mac = 4 * 512 * (257 + 512) + 2 * 4 * 512 * 1024 + 512 * 514
model.load_state_dict(sd, strict=True)
model = model.cuda().train()
model.train_arithmetic = ARITH
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
print(f"train step {which} B={B} {ARITH}: {dt * 1e3:.1f} ms, loss {loss.item():.5f}, ~{3 * 2 * mac * B * (T + 2) / dt / 1e12:.1f} TFLOP/s "
      f"({B * T / dt:.0f} frames/s), peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
