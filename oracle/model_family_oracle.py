"""CPU oracle for the sibling model families built from SequenceModel blocks:
Fast FullSubNet (recipes/dns_interspeech_2020/fast_fullsubnet/model.py) and the full-band baseline
(recipes/dns_interspeech_2020/fullband_baseline/model.py).

TEST INFRASTRUCTURE ONLY (same rule as fullsubnet_oracle.py): nothing under ``fullsubnet_amd/`` may
import this module.  numpy restatement, pinned on golden vectors made by running the reference
models themselves (tests/golden/make_golden_family.py).

Parity note: the reference takes Fast FullSubNet's mel filterbank from torchaudio (absent from the
reference tree and from this image).  ``melscale_fbanks`` restates torchaudio's documented
``melscale_fbanks(norm=None, mel_scale="htk")``; parity is UNPINNED at that boundary - the golden run
injects the filterbank through a stub module, and the forward tests treat ``mel_scale.fb`` as a
parameter so that everything downstream of it is pinned on the reference.
"""
from __future__ import annotations

import numpy as np

from . import fullsubnet_oracle as O
from fsn_synthetic import (IMPROVED_16K, IMPROVED_48K, IMPROVED_48K_769, make_fast_params, make_fullband_params,  # noqa: F401
                           make_improved_params)


def melscale_fbanks(n_freqs=257, f_min=0.0, f_max=8000.0, n_mels=64, sample_rate=16000):
    """Triangular HTK-mel filters without area normalisation -> [n_freqs, n_mels] (float64 math)."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = np.diff(f_pts)
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def sequence_block(x, params, prefix, num_layers, has_fc, activation, dtype=np.float32):
    """SequenceModel.forward (audio_zen/model/module/sequence_model.py:106-125), output layer optional
    (``output_size = 0``, :82-84)."""
    o = np.ascontiguousarray(np.asarray(x, dtype=dtype).transpose(0, 2, 1))
    p = f"{prefix}.sequence_model."
    for k in range(num_layers):
        o = O.lstm_layer(o, params[p + f"weight_ih_l{k}"], params[p + f"weight_hh_l{k}"],
                         params[p + f"bias_ih_l{k}"], params[p + f"bias_hh_l{k}"], dtype=dtype)
    if has_fc:
        w = np.asarray(params[f"{prefix}.fc_output_layer.weight"], dtype=dtype)
        b = np.asarray(params[f"{prefix}.fc_output_layer.bias"], dtype=dtype)
        o = (o @ w.T + b).astype(dtype)
    if activation == "ReLU":
        o = np.maximum(o, 0)
    elif activation:
        raise NotImplementedError(activation)
    return np.ascontiguousarray(o.transpose(0, 2, 1))


def real_time_downsampling(x, shrink_size):
    """fast_fullsubnet/model.py:108-129: frame 0, then block means of the remaining frames."""
    rest = x[..., 1:]
    blocks = [rest[..., i:i + shrink_size] for i in range(0, rest.shape[-1], shrink_size)]
    cols = [x[..., 0:1]] + [b.mean(axis=-1, keepdims=True, dtype=x.dtype) for b in blocks]
    return np.concatenate(cols, axis=-1)


def real_time_upsampling(x, shrink_size, target_len):
    """fast_fullsubnet/model.py:131-140."""
    return np.repeat(x, shrink_size, axis=-1)[..., :target_len]


def fast_fullsubnet_forward(mix_mag, params, look_ahead=2, shrink_size=2, num_mels=64,
                            noisy_input_num_neighbors=5, encoder_output_num_neighbors=0,
                            bottleneck_num_layers=2, dtype=np.float32):
    """fast_fullsubnet/model.py:143-202.  mix_mag [B, 1, F, T] -> [B, 2, F, T]."""
    x = np.asarray(mix_mag, dtype=dtype)
    x = np.pad(x, [(0, 0), (0, 0), (0, 0), (0, look_ahead)])
    B, C, F, T = x.shape
    fb = np.asarray(params["mel_scale.fb"], dtype=dtype)
    mel = (x.transpose(0, 1, 3, 2) @ fb).transpose(0, 1, 3, 2).astype(dtype)  # [B, 1, M, T]
    enc_in = O.offline_laplace_norm(mel, dtype).reshape(B, -1, T)
    e = sequence_block(enc_in, params, "encoder.0", 1, False, None, dtype)
    e = sequence_block(e, params, "encoder.1", 1, True, "ReLU", dtype)
    enc_out = e.reshape(B, C, -1, T)
    nu = O.freq_unfold(mel, noisy_input_num_neighbors).reshape(B, num_mels, 2 * noisy_input_num_neighbors + 1, T)
    eu = O.freq_unfold(enc_out, encoder_output_num_neighbors).reshape(B, num_mels,
                                                                      2 * encoder_output_num_neighbors + 1, T)
    bn_in = np.concatenate([nu, eu], axis=2)
    K = bn_in.shape[2]
    bn_s = O.offline_laplace_norm(real_time_downsampling(bn_in, shrink_size), dtype)
    bn_s = bn_s.reshape(B * num_mels, K, -1)
    bo = sequence_block(bn_s, params, "bottleneck", bottleneck_num_layers, True, "ReLU", dtype)
    bo = bo.reshape(B, num_mels, 1, -1).transpose(0, 2, 1, 3)
    bn_out = real_time_upsampling(bo, shrink_size, T)
    dec_in = np.concatenate([enc_out, bn_out], axis=2).reshape(B, -1, T)
    d = sequence_block(dec_in, params, "decoder_lstm.0", 1, False, None, dtype)
    d = sequence_block(d, params, "decoder_lstm.1", 1, True, None, dtype)
    return np.ascontiguousarray(d.reshape(B, 2, F, T)[..., look_ahead:])


def fullband_baseline_forward(noisy_mag, params, look_ahead=2, dtype=np.float32):
    """fullband_baseline/model.py:45-68."""
    x = np.asarray(noisy_mag, dtype=dtype)
    x = np.pad(x, [(0, 0), (0, 0), (0, 0), (0, look_ahead)])
    B, C, F, T = x.shape
    o = sequence_block(O.offline_laplace_norm(x, dtype).reshape(B, C * F, T), params, "fullband_model", 3, True, None,
                       dtype)
    return np.ascontiguousarray(o.reshape(B, 2, F, T)[..., look_ahead:])


# --------------------------------------------------------------------------- #
# Improved FullSubNet     recipes/dns_interspeech_2020/improved_fullsubnet/model.py
# --------------------------------------------------------------------------- #


def _norm_eps32(x, dtype=np.float32):
    """improved_fullsubnet/model.py:128-150: utterance-level mean, eps = fp32 epsilon (not 1e-5)."""
    x = np.asarray(x, dtype=dtype)
    mu = x.mean(axis=tuple(range(1, x.ndim)), keepdims=True, dtype=np.float64).astype(dtype)
    return (x / (mu + dtype(O.EPSILON))).astype(dtype)


def banded_unfold(x, lower, upper, centers, neighbors):
    """improved_fullsubnet/model.py:315-400.  x [B, 1, F, T] -> [B, N, 1, centers + 2 neighbors, T]: a window of
    that width slides over the band in steps of ``centers`` bins; bins beyond either end of the spectrum are
    mirrored (no edge repeat)."""
    B, C, F, T = x.shape
    assert C == 1 and (upper - lower) % centers == 0
    units = (upper - lower) // centers
    out = np.empty((B, units, 1, centers + 2 * neighbors, T), dtype=x.dtype)
    for u in range(units):
        for k in range(centers + 2 * neighbors):
            j = lower + u * centers + k - neighbors
            j = -j if j < 0 else j
            j = 2 * (F - 1) - j if j > F - 1 else j
            out[:, u, 0, k, :] = x[:, 0, j, :]
    return out


def improved_fullsubnet_forward(y, params, cfg, window, dtype=np.float32):
    """improved_fullsubnet/model.py:541-591: y [B, L] -> enhanced [B, 1, L]."""
    y = np.asarray(y, dtype=dtype)
    n_fft, hop = cfg["n_fft"], cfg["hop_length"]
    mag, _, re, im = O.stft(y, n_fft, hop, cfg["win_length"], window=window, dtype=dtype)
    x = (mag[:, None].astype(dtype) ** dtype(cfg["fdrc"])).astype(dtype)[:, :, :-1, :]
    B, _, F, T = x.shape
    fb = sequence_block(_norm_eps32(x, dtype).reshape(B, F, T), params, "fb_model", 2, True, None, dtype)
    fb = fb.reshape(B, 1, F, T)
    cuts = list(cfg["freq_cutoffs"])
    sections = []
    for i in range(len(cuts) + 1):
        lower = 0 if i == 0 else cuts[i - 1]
        upper = F if i == len(cuts) else cuts[i]
        a = banded_unfold(x, lower, upper, cfg["sb_num_center_freqs"][i], cfg["sb_num_neighbor_freqs"][i])
        b = banded_unfold(fb, lower, upper, cfg["fb_num_center_freqs"][i], cfg["fb_num_neighbor_freqs"][i])
        s = _norm_eps32(np.concatenate([a, b], axis=-2), dtype)
        N, K = s.shape[1], s.shape[3]
        o = sequence_block(s.reshape(B * N, K, T), params, f"sb_model.sb_models.{i}", 2, True, None, dtype)
        o = o.reshape(B, N, 2, -1, T).transpose(0, 2, 1, 3, 4).reshape(B, 2, -1, T)
        sections.append(o)
    crm = np.concatenate(sections, axis=-2)
    crm = np.pad(crm, [(0, 0), (0, 0), (0, 1), (0, 0)])
    out = O.istft((crm[:, 0] * re).astype(dtype), (crm[:, 1] * im).astype(dtype), n_fft, hop, cfg["win_length"],
                  length=y.shape[-1], window=window, dtype=dtype)
    return out[:, None, :]
