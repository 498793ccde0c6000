"""Generate golden vectors by running the REFERENCE itself (CPU, torch) in the authoring
container.  Run from the repo root:   python tests/golden/make_golden.py

The reference checkout (/root/reference) does not exist on the GPU box, so the outputs are
committed as small .npz fixtures next to this script.  Weights and inputs are produced by
``oracle.fullsubnet_oracle.make_params / make_noisy`` (numpy PCG64, platform independent) and
are therefore NOT stored, only their checksums are.

Reference entry points exercised (all imported unmodified from /root/reference):
  audio_zen.acoustics.feature.{stft, istft, drop_band}
  audio_zen.acoustics.mask.{decompress_cIRM, build_complex_ideal_ratio_mask, compress_cIRM}
  recipes/dns_interspeech_2020/fullsubnet/model.py:Model  (+ BaseModel norms / freq_unfold)
in the order of recipes/dns_interspeech_2020/inferencer.py:130-145.
"""
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.modules.setdefault("librosa", types.ModuleType("librosa"))  # feature.py:3, unused on the path
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/recipes/dns_interspeech_2020")

from audio_zen.acoustics.feature import drop_band, istft, stft  # noqa: E402
from audio_zen.acoustics.mask import (build_complex_ideal_ratio_mask, compress_cIRM,  # noqa: E402
                                      decompress_cIRM)
from fullsubnet.model import Model  # noqa: E402

from oracle.fullsubnet_oracle import make_noisy, make_params  # noqa: E402


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def build_model(params, norm_type, groups):
    m = Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0,
              sb_num_neighbors=15, fb_output_activate_function="ReLU",
              sb_output_activate_function=False, fb_model_hidden_size=512,
              sb_model_hidden_size=384, norm_type=norm_type, num_groups_in_drop_band=groups,
              weight_init=False).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m


def run_case(name, batch, length, norm_type, groups, gain, mask_gain=24.0, seed_w=0, seed_x=1234):
    params = make_params(seed=seed_w, gain=gain, mask_gain=mask_gain)
    noisy = make_noisy(batch, length, seed=seed_x)
    model = build_model(params, norm_type, groups)
    y = torch.from_numpy(noisy)
    with torch.no_grad():
        mag, phase, re, im = stft(y, 512, 256, 512)
        crm = model(mag.unsqueeze(1))  # [B, 2, F', T]
        out = dict(window=torch.hann_window(512).numpy(), mag=mag.numpy(), real=re.numpy(),
                   imag=im.numpy(), crm=crm.numpy())
        if groups == 1:
            p = crm.permute(0, 2, 3, 1)
            dm = decompress_cIRM(p)
            er = dm[..., 0] * re - dm[..., 1] * im
            ei = dm[..., 1] * re + dm[..., 0] * im
            enh = istft((er, ei), 512, 256, 512, length=y.size(-1), input_type="real_imag")
            out.update(dcrm=dm.numpy(), enh_real=er.numpy(), enh_imag=ei.numpy(), enhanced=enh.numpy())
            # intermediates for per-stage parity
            x = torch.nn.functional.pad(mag.unsqueeze(1), [0, 2])
            fb_in = model.norm(x).reshape(batch, 257, -1)
            fb_out = model.fb_model(fb_in)
            out.update(fb_output=fb_out.numpy())
    meta = dict(batch=batch, length=length, norm_type=norm_type, groups=groups, gain=gain, mask_gain=mask_gain,
                seed_w=seed_w, seed_x=seed_x, torch=torch.__version__,
                crc_noisy=crc(noisy), crc_w=crc(np.concatenate([v.ravel() for v in params.values()])))
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    c = out["crm"]
    print(f"{name}: crm range [{c.min():.3f}, {c.max():.3f}] |crm|>9.9: {(np.abs(c) > 9.9).mean():.4f} "
          f"size {os.path.getsize(path) / 1024:.0f} KiB")


def run_elementwise():
    """Known-answer vectors for the mask algebra and drop_band (mask.py, feature.py:309-345)."""
    rng = np.random.default_rng(7)
    m = np.concatenate([rng.uniform(-12, 12, 500), [9.9, -9.9, 9.899999, -9.899999, 0.0, 10.0, -10.0]]).astype(np.float32)
    raw = np.concatenate([rng.standard_normal(500) * 30, [-100.0, -100.5, -99.99, 0.0, 250.0]]).astype(np.float32)
    nr, ni, cr, ci = (rng.standard_normal((2, 5, 7)).astype(np.float32) for _ in range(4))
    x = rng.standard_normal((6, 3, 9, 4)).astype(np.float32)
    out = dict(
        m=m, dm=decompress_cIRM(torch.from_numpy(m)).numpy(),
        raw=raw, comp=compress_cIRM(torch.from_numpy(raw)).numpy(),
        nr=nr, ni=ni, cr=cr, ci=ci,
        cirm=build_complex_ideal_ratio_mask(*(torch.from_numpy(a) for a in (nr, ni, cr, ci))).numpy(),
        x=x, drop2=drop_band(torch.from_numpy(x), 2).numpy(), drop3=drop_band(torch.from_numpy(x), 3).numpy(),
    )
    np.savez_compressed(os.path.join(HERE, "elementwise.npz"), **out)
    print("elementwise: ok")


if __name__ == "__main__":
    torch.manual_seed(0)
    run_elementwise()
    run_case("fsn_offline_b2", batch=2, length=4096, norm_type="offline_laplace_norm", groups=1, gain=2.0)
    run_case("fsn_offline_b1_odd", batch=1, length=5003, norm_type="offline_laplace_norm", groups=1, gain=2.0, seed_x=99)
    run_case("fsn_cumulative_b2", batch=2, length=4096, norm_type="cumulative_laplace_norm", groups=1, gain=2.0)
    run_case("fsn_dropband_b4", batch=4, length=2560, norm_type="offline_laplace_norm", groups=2, gain=2.0)
