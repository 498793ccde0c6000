"""Golden vectors AT THE BASELINE SHAPES (3 s utterances: T = 188 frames at 16 kHz, T' = 190 recurrent steps),
produced by running the REFERENCE itself on CPU in the authoring container.  Run from the repo root:
    python tests/golden/make_golden_long.py

The short fixtures of make_golden.py / make_golden_family.py stop at 20 frames; these pin the 190-step
recurrences of BASELINE configs 2, 4 and 5 on the reference's own arithmetic (torch CPU: oneDNN LSTM, MKL FFT).
To keep the files small the frequency axis of the stored masks / spectra is sub-sampled at a fixed stride
(BIN_STRIDE, plus the last bin) and waveforms at SAMPLE_STRIDE; the strides are recorded in ``meta``.

  fsn_long_b2            recipes/dns_interspeech_2020/fullsubnet/model.py:Model in the order of inferencer.py:130-145
  fast_long_b2           fast_fullsubnet/model.py:Model                     (torchaudio.MelScale stubbed, see make_golden_family.py)
  improved_48k_long_b1   improved_fullsubnet/model.py:Model, 48 kHz example (:603-620), 3 s
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_family as FAM  # noqa: E402  (sets up the reference imports and the torchaudio stub)
import make_golden as G  # noqa: E402

from oracle import model_family_oracle as MF  # noqa: E402
from oracle.fullsubnet_oracle import make_noisy, make_params  # noqa: E402

BIN_STRIDE = 4
SAMPLE_STRIDE = 4


def bins(F):
    idx = list(range(0, F, BIN_STRIDE))
    if idx[-1] != F - 1:
        idx.append(F - 1)
    return np.asarray(idx)


def fsn_long(name="fsn_long_b2", batch=2, length=48000, seed_w=0, seed_x=1234, gain=2.0, mask_gain=24.0):
    params = make_params(seed=seed_w, gain=gain, mask_gain=mask_gain)
    noisy = make_noisy(batch, length, seed=seed_x)
    model = G.build_model(params, "offline_laplace_norm", 1)
    y = torch.from_numpy(noisy)
    with torch.no_grad():
        mag, _, re, im = G.stft(y, 512, 256, 512)
        crm = model(mag.unsqueeze(1))
        dm = G.decompress_cIRM(crm.permute(0, 2, 3, 1))
        er = dm[..., 0] * re - dm[..., 1] * im
        ei = dm[..., 1] * re + dm[..., 0] * im
        enh = G.istft((er, ei), 512, 256, 512, length=length, input_type="real_imag")
        x = torch.nn.functional.pad(mag.unsqueeze(1), [0, 2])
        fb_out = model.fb_model(model.norm(x).reshape(batch, 257, -1))
    b = bins(257)
    meta = dict(batch=batch, length=length, norm_type="offline_laplace_norm", groups=1, gain=gain, mask_gain=mask_gain,
                seed_w=seed_w, seed_x=seed_x, torch=torch.__version__, bin_stride=BIN_STRIDE,
                sample_stride=SAMPLE_STRIDE, crc_noisy=G.crc(noisy),
                crc_w=G.crc(np.concatenate([v.ravel() for v in params.values()])))
    frame_max = np.maximum(np.abs(re.numpy()), np.abs(im.numpy())).max(axis=1, keepdims=True)  # [B, 1, T], all bins
    out = dict(bins=b, real=re.numpy()[:, b], imag=im.numpy()[:, b], frame_max=frame_max, crm=crm.numpy()[:, :, b],
               fb_output=fb_out.numpy()[:, b], enhanced=enh.numpy()[:, ::SAMPLE_STRIDE],
               enhanced_absmax=np.float32(enh.abs().max().item()), meta=np.array(repr(meta)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    c = crm.numpy()
    print(f"{name}: T = {mag.shape[-1]} crm range [{c.min():.2f}, {c.max():.2f}] |crm| > 9.9: "
          f"{(np.abs(c) > 9.9).mean():.4f}  {os.path.getsize(path) / 1024:.0f} KiB")


def fast_long(name="fast_long_b2", batch=2, length=48000, seed_w=0, seed_x=77, gain=2.0):
    params = MF.make_fast_params(seed=seed_w, gain=gain)
    noisy = make_noisy(batch, length, seed=seed_x)
    m = FAM.FastModel(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
                      bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
                      encoder_output_num_neighbors=0, norm_type="offline_laplace_norm", weight_init=False).eval()
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        mag, _, _, _ = FAM.stft(torch.from_numpy(noisy), 512, 256, 512)
        crm = m(mag.unsqueeze(1))
    b = bins(257)
    meta = dict(batch=batch, length=length, seed_w=seed_w, seed_x=seed_x, gain=gain, torch=torch.__version__,
                bin_stride=BIN_STRIDE, crc_noisy=FAM.crc(noisy),
                crc_w=FAM.crc(np.concatenate([v.ravel() for v in params.values()])))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, bins=b, crm=crm.numpy()[:, :, b], fb=m.mel_scale.fb.numpy(), meta=np.array(repr(meta)))
    c = crm.numpy()
    print(f"{name}: crm {c.shape} range [{c.min():.2f}, {c.max():.2f}]  {os.path.getsize(path) / 1024:.0f} KiB")


def improved_long(name="improved_48k_long_b1", batch=1, length=144000, seed_w=1, seed_x=80):
    cfg = MF.IMPROVED_48K
    params = MF.make_improved_params(cfg, seed=seed_w)
    noisy = make_noisy(batch, length, seed=seed_x)
    m = FAM.ImprovedModel(**cfg).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    with torch.no_grad():
        enh = m(torch.from_numpy(noisy))
    meta = dict(batch=batch, length=length, seed_w=seed_w, seed_x=seed_x, torch=torch.__version__,
                sample_stride=8, crc_noisy=FAM.crc(noisy),
                crc_w=FAM.crc(np.concatenate([v.ravel() for v in params.values()])))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, enhanced=enh.numpy()[..., ::8], enhanced_absmax=np.float32(enh.abs().max().item()),
                        meta=np.array(repr(meta)))
    e = enh.numpy()
    print(f"{name}: enhanced {e.shape} rms {np.sqrt((e ** 2).mean()):.4f}  {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.manual_seed(0)
    fsn_long()
    fast_long()
    improved_long()
