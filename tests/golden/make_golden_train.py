"""Golden vectors of ONE TRAINING STEP produced by the REFERENCE itself (CPU, torch) - run in the
authoring container:   python tests/golden/make_golden_train.py

Follows recipes/dns_interspeech_2020/fullsubnet/trainer.py:41-71 with use_amp = false: reference
stft / build_complex_ideal_ratio_mask / drop_band / Model (nn.LSTM) / MSELoss / clip_grad_norm_(10)
/ Adam(lr 1e-3).  Stored: the loss, and for every parameter the clipped-gradient norm, a strided
sample of the gradient and of the updated parameter (the full tensors are 22 MB each).
Flags: --config3 / --config3x2 (BASELINE config 3 shapes), --amp-bf16 (the step under CPU autocast), --cumulative (the
shipped train_cumulativeLaplaceNorm.toml's norm), --amp-fp16 (the shipped use_amp = true arithmetic itself: torch.autocast("cpu",
float16) + GradScaler exactly as trainer.py:56-69; oneDNN has no fp16 LSTM primitive - "could not create a primitive descriptor
for the LSTM forward propagation primitive" - so that step runs with torch.backends.mkldnn.flags(enabled=False): ATen's own
LSTM cell, fp16 tensors, fp32 accumulation inside each product; ~25 min for the config-3 shape on 8 cores).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.modules.setdefault("librosa", types.ModuleType("librosa"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/recipes/dns_interspeech_2020")

from audio_zen.acoustics.feature import drop_band, stft  # noqa: E402
from audio_zen.acoustics.mask import build_complex_ideal_ratio_mask  # noqa: E402
from fullsubnet.model import Model  # noqa: E402

from oracle.fullsubnet_oracle import make_noisy, make_params  # noqa: E402

SAMPLE = 97  # stride of the per-parameter samples


def main(batch=4, length=2560, groups=2, name="fsn_train_b4", sample=SAMPLE, autocast=None, norm_type="offline_laplace_norm",
         scaler=None):
    params = make_params(seed=3)
    noisy = make_noisy(batch, length, seed=41)
    clean = 0.7 * make_noisy(batch, length, seed=42)
    model = Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                  fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                  sb_model_hidden_size=384, norm_type=norm_type, num_groups_in_drop_band=groups,
                  weight_init=False).train()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
    opt.zero_grad()
    noisy_mag, _, nr, ni = stft(torch.from_numpy(noisy), 512, 256, 512)
    _, _, cr, ci = stft(torch.from_numpy(clean), 512, 256, 512)
    cirm = build_complex_ideal_ratio_mask(nr, ni, cr, ci)
    cirm = drop_band(cirm.permute(0, 3, 1, 2), groups).permute(0, 2, 3, 1)
    # autocast: the reference's use_amp = true graph (trainer.py:56-62: model forward and loss inside the context, the
    # transforms and the target outside).  On the CPU the only 16-bit type nn.LSTM runs in is bfloat16 (oneDNN has no
    # fp16 LSTM primitive), and bf16 needs no loss scaling: GradScaler(enabled) would be the identity here.
    with torch.autocast("cpu", dtype=autocast, enabled=autocast is not None):
        crm = model(noisy_mag.unsqueeze(1)).permute(0, 2, 3, 1)
        loss = torch.nn.MSELoss()(cirm, crm)
    if scaler is None:
        loss.backward()
    else:  # trainer.py:63-69
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
    total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    out = dict(loss=np.float64(loss.item()), total_norm=np.float64(total_norm.item()))
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    if scaler is None:
        opt.step()
    else:
        out["scale_before"] = np.float64(scaler.get_scale())
        scaler.step(opt)
        scaler.update()
        out["scale_after"] = np.float64(scaler.get_scale())
    for k, p in model.named_parameters():
        out["gnorm/" + k] = np.float64(grads[k].norm().item())
        out["g/" + k] = grads[k].reshape(-1)[::sample].numpy().copy()
        out["p/" + k] = p.detach().reshape(-1)[::sample].numpy().copy()
    out["meta"] = np.array(repr(dict(batch=batch, length=length, groups=groups, seed_w=3, seed_noisy=41, seed_clean=42,
                                     clean_gain=0.7, sample=sample, torch=torch.__version__,
                                     autocast=str(autocast), norm_type=norm_type)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"loss {loss.item():.6f} total grad norm {total_norm.item():.4f} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--amp-fp16" in sys.argv:
        with torch.backends.mkldnn.flags(enabled=False):
            main(name="fsn_train_b4_f16", autocast=torch.float16, scaler=torch.amp.GradScaler("cpu"))
            if "--short" not in sys.argv:
                main(batch=16, length=49152, groups=2, name="fsn_train_c3_f16", sample=397, autocast=torch.float16,
                     scaler=torch.amp.GradScaler("cpu"))
    elif "--amp-bf16" in sys.argv:
        # the same two steps under torch.autocast("cpu", dtype=torch.bfloat16)
        main(name="fsn_train_b4_bf16", autocast=torch.bfloat16)
        main(batch=16, length=49152, groups=2, name="fsn_train_c3_bf16", sample=397, autocast=torch.bfloat16)
    elif "--cumulative" in sys.argv:
        # the other shipped training configuration (fullsubnet/train_cumulativeLaplaceNorm.toml:82): the same two steps with
        # norm_type = "cumulative_laplace_norm" (audio_zen/model/base_model.py:221-251; on the 4-D sub-band tensor every
        # unit is its own "sample") - the short batch and BASELINE config 3's per-rank shape
        main(name="fsn_train_cum_b4", norm_type="cumulative_laplace_norm")
        main(batch=16, length=49152, groups=2, name="fsn_train_cum_c3", sample=397, norm_type="cumulative_laplace_norm")
    elif "--config3x2" in sys.argv:
        # two ranks of BASELINE config 3 as ONE batch: 32 utterances x 3.072 s (what 2 x 16 under DistributedDataParallel
        # must reproduce: drop_band keeps the sample parity of the global batch when ranks take contiguous halves)
        main(batch=32, length=49152, groups=2, name="fsn_train_c3x2", sample=397)
    elif "--config3" in sys.argv:
        # BASELINE config 3 per-rank shape (fullsubnet/train.toml:46,92: 16 utterances x 3.072 s, drop_band groups 2)
        main(batch=16, length=49152, groups=2, name="fsn_train_c3", sample=397)
    else:
        main()
