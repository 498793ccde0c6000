"""Golden vectors for the FullSubNet constructor options no shipped TOML uses but the reference
accepts (recipes/dns_interspeech_2020/fullsubnet/model.py:10-70): the GRU branch of SequenceModel, the
three extra norms of BaseModel.norm_wrapper, fb_num_neighbors > 0 and other output activations -
produced by running the REFERENCE model on CPU.  Run from the repo root:
    python tests/golden/make_golden_variants.py [name ...]
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
sys.modules.setdefault("librosa", types.ModuleType("librosa"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/recipes/dns_interspeech_2020")

from audio_zen.acoustics.feature import stft  # noqa: E402
from fullsubnet.model import Model  # noqa: E402

from oracle.fullsubnet_oracle import make_noisy, make_params  # noqa: E402

# name -> constructor overrides (on top of the shipped FullSubNet arguments) + batch / groups
VARIANTS = {
    "var_gru_b2": dict(sequence_model="GRU"),
    "var_gaussian_b2": dict(norm_type="offline_gaussian_norm"),
    "var_cln_b2": dict(norm_type="cumulative_layer_norm"),
    "var_forgetting_b2": dict(norm_type="forgetting_norm"),
    "var_fbnn2_tanh_b3": dict(fb_num_neighbors=2, fb_output_activate_function="Tanh", batch=3, groups=2),
    # 33 x 257 = 8481 sub-band rows: more than two 16-row tiles per CU of an MI355X, where fullsubnet_amd runs the GRU on its
    # persistent many-row kernels (+ 19 left-over tiles step by step beside them); 9 frames keep the file small
    "var_gru_b33": dict(sequence_model="GRU", batch=33, length=2048),
}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def run(name, spec):
    spec = dict(spec)
    batch, groups, length = spec.pop("batch", 2), spec.pop("groups", 1), spec.pop("length", 4096)
    kw = dict(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
              fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
              sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=groups,
              weight_init=False)
    kw.update(spec)
    gates = 3 if kw["sequence_model"] == "GRU" else 4
    params = make_params(seed=2, gain=2.0, mask_gain=24.0, gates=gates, fb_num_neighbors=kw["fb_num_neighbors"])
    noisy = make_noisy(batch, length, seed=321)
    m = Model(**kw).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    with torch.no_grad():
        mag, _, _, _ = stft(torch.from_numpy(noisy), 512, 256, 512)
        crm = m(mag.unsqueeze(1)).numpy()
    meta = dict(kw=kw, batch=batch, length=length, seed_w=2, seed_x=321, gain=2.0, mask_gain=24.0, gates=gates,
                torch=torch.__version__, crc_noisy=crc(noisy),
                crc_w=crc(np.concatenate([v.ravel() for v in params.values()])))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), mag=mag.numpy(), crm=crm, meta=np.array(repr(meta)))
    print(f"{name}: crm {crm.shape} range [{crm.min():.3f}, {crm.max():.3f}]")


if __name__ == "__main__":
    for name, spec in VARIANTS.items():
        if len(sys.argv) < 2 or name in sys.argv[1:]:
            run(name, spec)
