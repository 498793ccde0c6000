"""Time the sibling model families on one MI355X (BASELINE configs 1 and 4):
python tools/bench_family.py [fast|fullband|improved16|improved48|improved769] [batch] [units=r/w]
units=r/w (improved* only): time what rank r of w computes under the frequency-axis shard (its share of every
section's units; the all-gather is not part of this single-GPU measurement)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fullsubnet_amd import decompress_cIRM, istft, stft  # noqa: E402
from fsn_synthetic import (IMPROVED_16K, IMPROVED_48K, IMPROVED_48K_769, make_fast_params, make_fullband_params,  # noqa: E402
                           make_improved_params, make_noisy)

which = sys.argv[1] if len(sys.argv) > 1 else "fast"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"fast": 256, "improved48": 32, "improved769": 32,
                                                "improved16": 32}.get(which, 1)
L = 48000
hop = 256
if which.startswith("improved"):
    from fullsubnet_amd.improved_fullsubnet import Model
    cfg = {"improved48": IMPROVED_48K, "improved769": IMPROVED_48K_769}.get(which, IMPROVED_16K)
    L = 48000 if which == "improved16" else 144000  # 3 s
    hop = cfg["hop_length"]
    model = Model(**cfg)
    sd = {k: torch.from_numpy(v) for k, v in make_improved_params(cfg, seed=3).items()}
    F = cfg["num_freqs"] - 1
    mmac = 4 * 512 * (F + 512) + 4 * 512 * 1024 + 512 * F  # full-band model
    cuts = [0] + list(cfg["freq_cutoffs"]) + [F]
    for i, c in enumerate(cfg["sb_num_center_freqs"]):
        units = (cuts[i + 1] - cuts[i]) // c
        k_in = 2 * (c + 30)
        mmac += units * (4 * 384 * (k_in + 384) + 4 * 384 * 768 + 384 * 2 * c)
elif which == "fast":
    from fullsubnet_amd.fast_fullsubnet import Model
    model = Model(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
                  bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
                  encoder_output_num_neighbors=0)
    sd = {k: torch.from_numpy(v) for k, v in make_fast_params(seed=3).items()}
    sd["mel_scale.fb"] = model.mel_scale.fb.clone()
    mmac = 62.9e6  # MAC / frame / utterance
else:
    from fullsubnet_amd.fullband_baseline import Model
    model = Model(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=None, look_ahead=2,
                  weight_init=False)
    sd = {k: torch.from_numpy(v) for k, v in make_fullband_params(seed=3).items()}
    mmac = 6.032384e6
model.load_state_dict(sd, strict=True)
model = model.cuda().eval()
shard = [a for a in sys.argv[3:] if a.startswith("units=")]
if shard:
    r, w = (int(v) for v in shard[0][6:].split("/"))
    sb = model.sb_model

    def one_rank(noisy_mag, fb_output, unit_group=None):
        local = sb.forward_units(noisy_mag, fb_output, r, w)
        # stand-in for the gathered result (timing only): this rank's units repeated to the full count
        full = [t.repeat((n + max(t.shape[0], 1) - 1) // max(t.shape[0], 1), 1, 1, 1, 1)[:n] if t.shape[0]
                else t.new_zeros((n,) + tuple(t.shape[1:]))
                for t, n in zip(local, sb.num_units(noisy_mag.size(2)))]
        return sb.assemble_units(full)

    sb.forward = one_rank
    which_label = f" units {r}/{w}"
else:
    which_label = ""
noisy = torch.from_numpy(make_noisy(min(B, 8), L, seed=1)).cuda().repeat((B + 7) // 8, 1)[:B].contiguous()


@torch.no_grad()
def enhance(y):
    if which.startswith("improved"):
        return model(y)  # waveform in, waveform out
    mag, _, re, im = stft(y, 512, 256, 512)
    crm = decompress_cIRM(model(mag.unsqueeze(1)).permute(0, 2, 3, 1))
    er = crm[..., 0] * re - crm[..., 1] * im
    ei = crm[..., 1] * re + crm[..., 0] * im
    return istft((er, ei), 512, 256, 512, length=y.size(-1), input_type="real_imag")


for _ in range(2):
    out = enhance(noisy)
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K):
    out = enhance(noisy)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
T = 1 + L // hop
la = 0 if which.startswith("improved") else 2
sr = 48000 if which in ("improved48", "improved769") else 16000
print(f"{which}{which_label} B={B}: {dt * 1e3:.2f} ms / batch, {B * T / dt:.0f} frames/s ({B * L / sr / dt:.0f} x real time), "
      f"~{2 * mmac * B * (T + la) / dt / 1e12:.1f} TFLOP/s, finite={bool(torch.isfinite(out).all())}")
