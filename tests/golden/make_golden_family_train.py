"""Golden vectors of ONE TRAINING STEP of the two sibling recipes that ship training TOMLs, produced by the REFERENCE itself
(CPU, torch, fp32) in the authoring container:   python tests/golden/make_golden_family_train.py

  fast_fullsubnet/train_shrinkSize2.toml:69-79 + fast_fullsubnet/trainer.py:33-76     -> fast_train_b3.npz
  fullband_baseline/train.toml:70-78 + fullband_baseline/trainer.py:32-71             -> fullband_train_b3.npz

Both trainers: reference stft / build_complex_ideal_ratio_mask (no drop_band in these recipes) / Model / MSELoss /
clip_grad_norm_(10) / Adam(lr 1e-3, betas 0.9 0.999), here with use_amp = false.  Stored like make_golden_train.py: the loss, the
total gradient norm, per parameter the clipped-gradient norm and strided samples of the gradient and the updated parameter.
Fast FullSubNet takes its mel filterbank from torchaudio, absent here: the MelScale stub of make_golden_family.py (a restatement of
torchaudio's documented HTK filterbank: parity unpinned AT THAT BOUNDARY, pinned on the reference's code behind it); the
filterbank is a buffer, not a parameter - it has no gradient.

  --b24: the bottleneck on 24 x 64 = 1536 rows (the smallest batch whose bottleneck is ONE piece of whole clusters for the persistent
  training kernels: fullsubnet_amd.train.lstm2_train_chunks), 12 288 samples: fast_train_b24.npz (fp32) and fast_train_b24_f16.npz =
  the SHIPPED arithmetic (train_shrinkSize2.toml:5 use_amp = true; fast_fullsubnet/trainer.py:52-66): torch.autocast("cpu", float16)
  + GradScaler, with torch.backends.mkldnn.flags(enabled=False) (oneDNN has no fp16 LSTM primitive: make_golden_train.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_family as G  # noqa: E402  (installs the torchaudio / torchinfo / librosa stubs, imports the reference models)
from audio_zen.acoustics.mask import build_complex_ideal_ratio_mask  # noqa: E402

SAMPLE = 211


def step(model, params, name, batch, length, meta, autocast=None, scaler=None):
    noisy = G.make_noisy(batch, length, seed=41)
    clean = 0.7 * G.make_noisy(batch, length, seed=42)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
    opt.zero_grad()
    noisy_mag, _, nr, ni = G.stft(torch.from_numpy(noisy), 512, 256, 512)
    _, _, cr, ci = G.stft(torch.from_numpy(clean), 512, 256, 512)
    cirm = build_complex_ideal_ratio_mask(nr, ni, cr, ci)
    with torch.autocast("cpu", dtype=autocast, enabled=autocast is not None):  # fast_fullsubnet/trainer.py:52-57
        crm = model(noisy_mag.unsqueeze(1)).permute(0, 2, 3, 1)
        loss = torch.nn.MSELoss()(cirm, crm)
    if scaler is None:
        loss.backward()
    else:  # trainer.py:59-66
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
    total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    out = dict(loss=np.float64(loss.item()), total_norm=np.float64(total_norm.item()))
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    if scaler is None:
        opt.step()
    else:
        scaler.step(opt)
        scaler.update()
    for k, p in model.named_parameters():
        out["gnorm/" + k] = np.float64(grads[k].norm().item())
        out["g/" + k] = grads[k].reshape(-1)[::SAMPLE].numpy().copy()
        out["p/" + k] = p.detach().reshape(-1)[::SAMPLE].numpy().copy()
    out["meta"] = np.array(repr(dict(batch=batch, length=length, seed_noisy=41, seed_clean=42, clean_gain=0.7, sample=SAMPLE,
                                     torch=torch.__version__, autocast=str(autocast), **meta)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss {loss.item():.6f} total grad norm {total_norm.item():.4f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def fast_model():
    params = G.MF.make_fast_params(seed=3, gain=1.0)
    m = G.FastModel(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
                    bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
                    encoder_output_num_neighbors=0, norm_type="offline_laplace_norm", weight_init=False)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    m.load_state_dict(sd, strict=True)
    return m, params


def main():
    torch.manual_seed(0)
    if "--b24" in sys.argv:
        m, params = fast_model()
        step(m, params, "fast_train_b24", 24, 12288, dict(seed_w=3, gain=1.0, model="fast_fullsubnet"))
        with torch.backends.mkldnn.flags(enabled=False):
            m, params = fast_model()
            step(m, params, "fast_train_b24_f16", 24, 12288, dict(seed_w=3, gain=1.0, model="fast_fullsubnet"),
                 autocast=torch.float16, scaler=torch.amp.GradScaler("cpu"))
        return
    m, params = fast_model()
    step(m, params, "fast_train_b3", 3, 6144, dict(seed_w=3, gain=1.0, model="fast_fullsubnet"))
    params = G.MF.make_fullband_params(seed=3, gain=1.0, out_gain=2.0)
    m = G.FullbandModel(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=False, look_ahead=2,
                        norm_type="offline_laplace_norm", weight_init=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    step(m, params, "fullband_train_b3", 3, 6144, dict(seed_w=3, gain=1.0, out_gain=2.0, model="fullband_baseline"))


if __name__ == "__main__":
    main()
