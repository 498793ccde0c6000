"""Golden vectors for the sibling model families, produced by running the REFERENCE models on CPU in
the authoring container.  Run from the repo root:   python tests/golden/make_golden_family.py

  recipes/dns_interspeech_2020/fast_fullsubnet/model.py:Model      (BASELINE config 4)
  recipes/dns_interspeech_2020/fullband_baseline/model.py:Model    (BASELINE config 1)

fast_fullsubnet/model.py imports torchaudio (:3) and torchinfo (:5), which are neither in the
reference tree nor in this image.  torchinfo is only used by the module's __main__ block; of
torchaudio only ``transforms.MelScale`` is used (:57-63).  Both are stubbed here; the MelScale stub
applies the filterbank of oracle.model_family_oracle.melscale_fbanks (a restatement of torchaudio's
documented HTK filterbank - parity unpinned at that boundary) and the filterbank itself is stored in
the fixture, so that everything downstream of the mel matmul is pinned on the reference's own code.
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
from oracle import model_family_oracle as MF  # noqa: E402
from oracle.fullsubnet_oracle import make_noisy  # noqa: E402


class _MelScale(torch.nn.Module):
    def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft):
        super().__init__()
        fb = MF.melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate).astype(np.float32)
        self.register_buffer("fb", torch.from_numpy(fb))

    def forward(self, specgram):
        return torch.matmul(specgram.transpose(-1, -2), self.fb).transpose(-1, -2)


ta = types.ModuleType("torchaudio")
ta.transforms = types.ModuleType("torchaudio.transforms")
ta.transforms.MelScale = _MelScale
ti = types.ModuleType("torchinfo")
ti.summary = lambda *a, **k: None
sys.modules.update({"torchaudio": ta, "torchaudio.transforms": ta.transforms, "torchinfo": ti})
sys.modules.setdefault("librosa", types.ModuleType("librosa"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/recipes/dns_interspeech_2020")

from audio_zen.acoustics.feature import istft, stft  # noqa: E402
from fast_fullsubnet.model import Model as FastModel  # noqa: E402
from fullband_baseline.model import Model as FullbandModel  # noqa: E402
from improved_fullsubnet.model import Model as ImprovedModel  # noqa: E402


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def save(name, out, meta):
    out["meta"] = np.array(repr(meta))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    c = out["crm"]
    print(f"{name}: crm {c.shape} range [{c.min():.3f}, {c.max():.3f}]")


def fast_case(name, batch, length, seed_w=0, seed_x=77, gain=2.0):
    params = MF.make_fast_params(seed=seed_w, gain=gain)
    noisy = make_noisy(batch, length, seed=seed_x)
    m = FastModel(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
                  bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
                  encoder_output_num_neighbors=0, norm_type="offline_laplace_norm", weight_init=False).eval()
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        mag, _, _, _ = stft(torch.from_numpy(noisy), 512, 256, 512)
        crm = m(mag.unsqueeze(1))
    meta = dict(batch=batch, length=length, seed_w=seed_w, seed_x=seed_x, gain=gain, torch=torch.__version__,
                crc_noisy=crc(noisy), crc_w=crc(np.concatenate([v.ravel() for v in params.values()])))
    save(name, dict(mag=mag.numpy(), crm=crm.numpy(), fb=m.mel_scale.fb.numpy()), meta)


def fullband_case(name, batch, length, seed_w=0, seed_x=78, gain=1.5):
    params = MF.make_fullband_params(seed=seed_w, gain=gain)
    noisy = make_noisy(batch, length, seed=seed_x)
    m = FullbandModel(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=None,
                      look_ahead=2, norm_type="offline_laplace_norm", weight_init=False).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    with torch.no_grad():
        mag, _, _, _ = stft(torch.from_numpy(noisy), 512, 256, 512)
        crm = m(mag.unsqueeze(1))
    meta = dict(batch=batch, length=length, seed_w=seed_w, seed_x=seed_x, gain=gain, torch=torch.__version__,
                crc_noisy=crc(noisy), crc_w=crc(np.concatenate([v.ravel() for v in params.values()])))
    save(name, dict(mag=mag.numpy(), crm=crm.numpy()), meta)


def improved_case(name, cfg, batch, length, seed_w=0, seed_x=80):
    """improved_fullsubnet/model.py:Model, waveform in -> waveform out."""
    params = MF.make_improved_params(cfg, seed=seed_w)
    noisy = make_noisy(batch, length, seed=seed_x)
    m = ImprovedModel(**cfg).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    with torch.no_grad():
        enh = m(torch.from_numpy(noisy))
    meta = dict(batch=batch, length=length, seed_w=seed_w, seed_x=seed_x, torch=torch.__version__,
                crc_noisy=crc(noisy), crc_w=crc(np.concatenate([v.ravel() for v in params.values()])))
    out = dict(enhanced=enh.numpy(), window=torch.hann_window(cfg["win_length"]).numpy(), meta=np.array(repr(meta)))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    e = out["enhanced"]
    print(f"{name}: enhanced {e.shape} rms {np.sqrt((e ** 2).mean()):.4f} (input rms {np.sqrt((noisy ** 2).mean()):.4f})")


STFT_SHAPES = [(512, 128), (960, 480), (400, 100), (1536, 384)]  # (n_fft, hop); 512/256 is in make_golden.py


def stft_generic_case(name, batch=2, length=5000, seed_x=79):
    """audio_zen/acoustics/feature.py stft / istft (= torch.stft / torch.istft) at the transform shapes
    of the other recipes (improved_fullsubnet/model.py:603-620) and two odd ones."""
    noisy = make_noisy(batch, length, seed=seed_x)
    out = {}
    for n_fft, hop in STFT_SHAPES:
        mag, _, re, im = stft(torch.from_numpy(noisy), n_fft, hop, n_fft)
        back = istft((re * 0.5, im * 0.5 + re * 0.25), n_fft, hop, n_fft, length=length, input_type="real_imag")
        k = f"{n_fft}_{hop}"
        out.update({f"win/{k}": torch.hann_window(n_fft).numpy(), f"re/{k}": re.numpy(), f"im/{k}": im.numpy(),
                    f"mag/{k}": mag.numpy(), f"back/{k}": back.numpy()})
    out["meta"] = np.array(repr(dict(batch=batch, length=length, seed_x=seed_x, torch=torch.__version__,
                                     crc_noisy=crc(noisy))))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if k.startswith("re/")})


CASES = {
    "stft_generic": lambda: stft_generic_case("stft_generic"),
    "improved_16k_b2": lambda: improved_case("improved_16k_b2", MF.IMPROVED_16K, 2, 4000),
    "improved_48k_b2": lambda: improved_case("improved_48k_b2", MF.IMPROVED_48K, 2, 9600, seed_w=1),
    # BASELINE config 5's "769 bins" (n_fft 1536 / hop 768): the reference model accepts these hyper-parameters
    # (improved_fullsubnet/model.py:315-400, 541-591); sections of 32 + 40 + 12 + 6 units, 22 frames
    "improved_769_b2": lambda: improved_case("improved_769_b2", MF.IMPROVED_48K_769, 2, 16000, seed_w=2, seed_x=81),
    "fast_b2_even": lambda: fast_case("fast_b2_even", 2, 8192),  # T' = 35: 34 frames after the first -> all blocks full
    "fast_b3_odd": lambda: fast_case("fast_b3_odd", 3, 8192 - 256, seed_w=1),  # T' = 34: 33 frames -> last block of 1
    "fullband_b2": lambda: fullband_case("fullband_b2", 2, 8192),
}

if __name__ == "__main__":
    for case in (sys.argv[1:] or list(CASES)):  # python tests/golden/make_golden_family.py [case ...]
        CASES[case]()
