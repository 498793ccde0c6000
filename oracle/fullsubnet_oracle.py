"""CPU oracle for the FullSubNet enhancement path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fullsubnet_amd/`` may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the *checker* of the HIP path.

This is a numpy restatement (explicit loops over time, explicit gate algebra,
explicit index maps) of what the reference computes on the path

    stft -> |X| -> look-ahead pad -> norm -> full-band LSTM -> sub-band unfold
         -> norm -> sub-band LSTM -> cIRM decompress -> complex mask -> istft

The arithmetic of the reference lives in ATen (torch.stft / nn.LSTM / F.unfold,
PyTorch 2.10 in this image); the restatement below is written against the
reference *call sites*, cited per function as ``path:line`` relative to the
reference checkout.  It is pinned in ``tests/test_oracle_golden.py`` against
golden vectors produced by running the reference itself
(``tests/golden/make_golden.py``), so parity is pinned, not assumed.

Every function takes ``dtype`` (np.float32 mirrors the reference's precision;
np.float64 is the arbiter used when the GPU and oneDNN disagree in the last
bits).
"""
from __future__ import annotations

import numpy as np

EPSILON = np.finfo(np.float32).eps  # audio_zen/constant.py:9


# --------------------------------------------------------------------------- #
# STFT / iSTFT                                  audio_zen/acoustics/feature.py
# --------------------------------------------------------------------------- #
def hann_window(n_fft: int, dtype=np.float32) -> np.ndarray:
    """Periodic Hann window, ``torch.hann_window(n_fft)`` (feature.py:38,89).

    torch evaluates ``0.5 - 0.5*cos(2*pi*n/N)`` with an fp32 SLEEF cosine, which
    numpy cannot reproduce bit-for-bit (1 ULP on ~10 % of taps), so every
    function below also accepts the window as an argument; the golden fixture
    carries torch's own window.
    """
    n = np.arange(n_fft, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(dtype)


def _reflect_pad(y: np.ndarray, pad: int) -> np.ndarray:
    # torch.stft(center=True, pad_mode="reflect"): no edge repeat.
    return np.pad(y, [(0, 0), (pad, pad)], mode="reflect")


def stft(y, n_fft=512, hop_length=256, win_length=512, window=None, dtype=np.float32):
    """feature.py:9-50 -> torch.stft(y, n_fft, hop, win, hann, return_complex).

    Defaults center=True, pad_mode="reflect", normalized=False, onesided=True.
    The frame*window product is rounded to ``dtype`` (as ATen does) and the DFT
    itself is evaluated in float64, i.e. this is the exactly-rounded value that
    MKL's fp32 FFT approximates to a few ULP.

    Returns (mag, phase, real, imag), each [B, F, T].
    """
    assert win_length == n_fft, "every shipped TOML uses win_length == n_fft"
    y = np.asarray(y, dtype=dtype)
    assert y.ndim == 2
    if window is None:
        window = hann_window(n_fft, dtype)
    window = np.asarray(window, dtype=dtype)
    yp = _reflect_pad(y, n_fft // 2)
    n_frames = 1 + (yp.shape[1] - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    frames = (yp[:, idx] * window[None, None, :]).astype(dtype)  # [B, T, n_fft]
    spec = np.fft.rfft(frames.astype(np.float64), axis=-1)  # [B, T, F]
    spec = np.transpose(spec, (0, 2, 1))  # [B, F, T]
    real = spec.real.astype(dtype)
    imag = spec.imag.astype(dtype)
    mag = np.sqrt(real.astype(np.float64) ** 2 + imag.astype(np.float64) ** 2).astype(dtype)
    phase = np.arctan2(imag, real).astype(dtype)
    return mag, phase, real, imag


def istft(real, imag, n_fft=512, hop_length=256, win_length=512, length=None, window=None,
          dtype=np.float32):
    """feature.py:53-91 with input_type="real_imag" -> torch.istft(..., length=).

    irfft per frame (imaginary parts of the DC and Nyquist bins are ignored,
    as every C2R transform does), multiply by the window, overlap-add, divide
    by the overlap-added squared window, drop the n_fft//2 centre padding and
    trim / zero-pad to ``length``.
    """
    assert win_length == n_fft
    real = np.asarray(real, dtype=dtype)
    imag = np.asarray(imag, dtype=dtype)
    if window is None:
        window = hann_window(n_fft, dtype)
    window = np.asarray(window, dtype=dtype)
    B, F, T = real.shape
    spec = (real.astype(np.float64) + 1j * imag.astype(np.float64)).transpose(0, 2, 1)  # [B,T,F]
    frames = np.fft.irfft(spec, n=n_fft, axis=-1).astype(dtype)  # [B, T, n_fft]
    frames = (frames * window[None, None, :]).astype(dtype)
    total = n_fft + hop_length * (T - 1)
    y = np.zeros((B, total), dtype=dtype)
    env = np.zeros((total,), dtype=dtype)
    wsq = (window * window).astype(dtype)
    for t in range(T):
        y[:, t * hop_length: t * hop_length + n_fft] += frames[:, t]
        env[t * hop_length: t * hop_length + n_fft] += wsq
    start = n_fft // 2
    end = total - start if length is None else start + length
    y = y[:, start:end]
    env = env[start:end]
    y = (y / env[None, : y.shape[1]]).astype(dtype)
    if length is not None and y.shape[1] < length:
        y = np.pad(y, [(0, 0), (0, length - y.shape[1])])
    return y


def mag_phase(real, imag, dtype=np.float32):
    """feature.py:94-96."""
    r = np.asarray(real, np.float64)
    i = np.asarray(imag, np.float64)
    return np.sqrt(r * r + i * i).astype(dtype), np.arctan2(i, r).astype(dtype)


def drop_band(x, num_groups=2):
    """feature.py:309-345 (dup base_model.py:254-292).  x: [B, C, F, T]."""
    B, _, F, _ = x.shape
    assert B > num_groups
    if num_groups <= 1:
        return x
    if F % num_groups != 0:
        x = x[:, :, : F - F % num_groups, :]
        F = x.shape[2]
    out = []
    for g in range(num_groups):
        out.append(x[g::num_groups][:, :, g:F:num_groups, :])
    return np.concatenate(out, axis=0)


# --------------------------------------------------------------------------- #
# cIRM mask math                                   audio_zen/acoustics/mask.py
# --------------------------------------------------------------------------- #
def compress_cIRM(mask, K=10, C=0.1, dtype=np.float32):
    """mask.py:32-44."""
    mask = np.asarray(mask, dtype=dtype)
    mask = (dtype(-100) * (mask <= -100) + mask * (mask > -100)).astype(dtype)
    e = np.exp((dtype(-C) * mask).astype(dtype)).astype(dtype)
    return (dtype(K) * (dtype(1) - e) / (dtype(1) + e)).astype(dtype)


def build_complex_ideal_ratio_mask(noisy_real, noisy_imag, clean_real, clean_imag, dtype=np.float32):
    """mask.py:7-29.  Returns [B, F, T, 2] (compressed)."""
    nr, ni, cr, ci = (np.asarray(a, dtype=dtype) for a in (noisy_real, noisy_imag, clean_real, clean_imag))
    den = (nr * nr + ni * ni + dtype(EPSILON)).astype(dtype)
    mr = ((nr * cr + ni * ci) / den).astype(dtype)
    mi = ((nr * ci - ni * cr) / den).astype(dtype)
    return compress_cIRM(np.stack((mr, mi), axis=-1), K=10, C=0.1, dtype=dtype)


def decompress_cIRM(mask, K=10, limit=9.9, dtype=np.float32):
    """mask.py:47-64."""
    mask = np.asarray(mask, dtype=dtype)
    lim = dtype(limit)
    mask = (lim * (mask >= lim) - lim * (mask <= -lim) + mask * (np.abs(mask) < lim)).astype(dtype)
    return (dtype(-K) * np.log(((dtype(K) - mask) / (dtype(K) + mask)).astype(dtype))).astype(dtype)


def complex_mul(noisy_r, noisy_i, mask_r, mask_i):
    """mask.py:67-70 == inferencer.py:139-140."""
    r = noisy_r * mask_r - noisy_i * mask_i
    i = noisy_r * mask_i + noisy_i * mask_r
    return r, i


# --------------------------------------------------------------------------- #
# norms + unfold                                audio_zen/model/base_model.py
# --------------------------------------------------------------------------- #
def offline_laplace_norm(x, dtype=np.float32):
    """base_model.py:204-218.  One mean per sample over every other dim; eps 1e-5."""
    x = np.asarray(x, dtype=dtype)
    mu = x.mean(axis=tuple(range(1, x.ndim)), keepdims=True, dtype=np.float64).astype(dtype)
    return (x / (mu + dtype(1e-5))).astype(dtype)


def cumulative_laplace_norm(x, dtype=np.float32):
    """base_model.py:221-251.  x: [B, C, F, T]; dim 1 is folded into batch (Q4)."""
    x = np.asarray(x, dtype=dtype)
    B, C, F, T = x.shape
    xr = x.reshape(B * C, F, T)
    step_sum = xr.sum(axis=1, dtype=np.float64)
    cum = np.cumsum(step_sum, axis=-1)
    count = np.arange(F, F * T + 1, F, dtype=np.float64)[None, :]
    mean = (cum / count).astype(dtype)[:, None, :]
    return (xr / (mean + dtype(EPSILON))).astype(dtype).reshape(B, C, F, T)


def offline_gaussian_norm(x, dtype=np.float32):
    """base_model.py:295-310.  (x - mean) / (std + 1e-5), torch.std = unbiased (N - 1)."""
    x = np.asarray(x, dtype=dtype)
    ax = tuple(range(1, x.ndim))
    mu = x.mean(axis=ax, keepdims=True, dtype=np.float64)
    std = np.sqrt(((x.astype(np.float64) - mu) ** 2).sum(axis=ax, keepdims=True) / (x[0].size - 1))
    return ((x - mu.astype(dtype)) / (std.astype(dtype) + dtype(1e-5))).astype(dtype)


def cumulative_layer_norm(x, dtype=np.float32):
    """base_model.py:312-354.  Running mean / variance over (F, frames <= t); dim 1 folded into batch."""
    x = np.asarray(x, dtype=dtype)
    B, C, F, T = x.shape
    xr = x.reshape(B * C, F, T)
    s1 = np.cumsum(xr.sum(axis=1, dtype=np.float64), axis=-1)
    s2 = np.cumsum((xr.astype(np.float64) ** 2).sum(axis=1), axis=-1)
    count = np.arange(F, F * T + 1, F, dtype=np.float64)[None, :]
    mean = s1 / count
    var = (s2 - 2 * mean * s1) / count + mean ** 2
    std = np.sqrt(var + EPSILON)
    out = (xr - mean.astype(dtype)[:, None, :]) / std.astype(dtype)[:, None, :]
    return out.astype(dtype).reshape(B, C, F, T)


def forgetting_norm(x, sample_length=192, dtype=np.float32):
    """base_model.py:103-151.  mu_t = a_t mu_{t-1} + (1 - a_t) mean(x[:, :, t]) with
    a_t = min((t - 1) / (t + 1), alpha) while t < sample_length (so a_0 = -1: the reference's own
    start-up, mu_0 = 2 mean_0), alpha = (L - 1) / (L + 1) afterwards; x / (mu + 1e-10)."""
    x = np.asarray(x, dtype=dtype)
    B, C, F, T = x.shape
    xr = x.reshape(B, C * F, T)
    frame_mean = xr.mean(axis=1, dtype=np.float64).astype(dtype)  # [B, T]
    alpha = (sample_length - 1) / (sample_length + 1)
    mu = np.zeros((B,), dtype=dtype)
    mus = np.empty((B, T), dtype=dtype)
    for t in range(T):
        a = dtype(min((t - 1) / (t + 1), alpha)) if t < sample_length else dtype(alpha)
        mu = (a * mu + (dtype(1) - a) * frame_mean[:, t]).astype(dtype)
        mus[:, t] = mu
    return (xr / (mus[:, None, :] + dtype(1e-10))).astype(dtype).reshape(B, C, F, T)


NORMS = {"offline_laplace_norm": offline_laplace_norm, "cumulative_laplace_norm": cumulative_laplace_norm,
         "offline_gaussian_norm": offline_gaussian_norm, "cumulative_layer_norm": cumulative_layer_norm,
         "forgetting_norm": forgetting_norm}


def reflect_index(j: np.ndarray, F: int) -> np.ndarray:
    """F.pad(mode="reflect") source index (no edge repeat) for positions j in [-N, F+N)."""
    j = np.where(j < 0, -j, j)
    return np.where(j >= F, 2 * (F - 1) - j, j)


def freq_unfold(x, num_neighbors):
    """base_model.py:14-46.  x: [B, C, F, T] -> [B, F, C, 2N+1, T].

    Row k of sub-band unit f is bin reflect(f + k - N) of the input.
    """
    B, C, F, T = x.shape
    if num_neighbors <= 0:
        return x.transpose(0, 2, 1, 3).reshape(B, F, C, 1, T)
    k = np.arange(2 * num_neighbors + 1)
    src = reflect_index(np.arange(F)[:, None] + k[None, :] - num_neighbors, F)  # [F, 2N+1]
    out = x[:, :, src, :]  # [B, C, F, 2N+1, T]
    return np.ascontiguousarray(out.transpose(0, 2, 1, 3, 4))


# --------------------------------------------------------------------------- #
# SequenceModel              audio_zen/model/module/sequence_model.py:26-125
# --------------------------------------------------------------------------- #
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


try:  # BLAS backend for the two GEMMs of the LSTM cell: torch's CPU matmul (MKL, all host cores) is
    # ~8x faster than this image's numpy/OpenBLAS build; everything else stays explicit numpy.
    import torch as _torch

    def _matmul(a, b):
        return (_torch.from_numpy(np.ascontiguousarray(a)) @ _torch.from_numpy(np.ascontiguousarray(b))).numpy()
except ImportError:  # pragma: no cover
    def _matmul(a, b):
        return a @ b


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, dtype=np.float32):
    """One unidirectional nn.LSTM layer, batch_first, h0 = c0 = 0.

    x: [N, T, I].  PyTorch gate order along the 4H axis is (i, f, g, o):
        i = sigmoid(W_ii x + b_ii + W_hi h + b_hi)      f, o likewise
        g = tanh(W_ig x + b_ig + W_hg h + b_hg)
        c = f * c + i * g ;  h = o * tanh(c)
    Returns the hidden sequence [N, T, H].
    """
    x = np.asarray(x, dtype=dtype)
    w_ih = np.asarray(w_ih, dtype=dtype)
    w_hh = np.asarray(w_hh, dtype=dtype)
    bias = (np.asarray(b_ih, dtype=dtype) + np.asarray(b_hh, dtype=dtype)).astype(dtype)
    N, T, I = x.shape
    H = w_hh.shape[1]
    h = np.zeros((N, H), dtype=dtype)
    c = np.zeros((N, H), dtype=dtype)
    out = np.empty((T, N, H), dtype=dtype)  # time-major scratch: every per-step slice is contiguous
    w_ih_t = np.ascontiguousarray(w_ih.T)
    w_hh_t = np.ascontiguousarray(w_hh.T)
    xw = _matmul(np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(T * N, I), w_ih_t)
    xw += bias
    xw = xw.reshape(T, N, 4 * H)
    for t in range(T):
        gates = xw[t]
        gates += _matmul(h, w_hh_t)
        i = _sigmoid(gates[:, 0:H])
        f = _sigmoid(gates[:, H:2 * H])
        g = np.tanh(gates[:, 2 * H:3 * H])
        o = _sigmoid(gates[:, 3 * H:4 * H])
        c = f * c + i * g
        h = o * np.tanh(c)
        out[t] = h
    return np.ascontiguousarray(out.transpose(1, 0, 2))


def gru_layer(x, w_ih, w_hh, b_ih, b_hh, dtype=np.float32):
    """One unidirectional nn.GRU layer, batch_first, h0 = 0 (sequence_model.py:59-66).

    x: [N, T, I].  PyTorch gate order along the 3H axis is (r, z, n):
        r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)      z likewise
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn))
        h = n + z * (h - n)
    """
    x = np.asarray(x, dtype=dtype)
    w_ih = np.asarray(w_ih, dtype=dtype)
    w_hh = np.asarray(w_hh, dtype=dtype)
    b_ih = np.asarray(b_ih, dtype=dtype)
    b_hh = np.asarray(b_hh, dtype=dtype)
    N, T, I = x.shape
    H = w_hh.shape[1]
    h = np.zeros((N, H), dtype=dtype)
    out = np.empty((T, N, H), dtype=dtype)
    w_ih_t = np.ascontiguousarray(w_ih.T)
    w_hh_t = np.ascontiguousarray(w_hh.T)
    xw = _matmul(np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(T * N, I), w_ih_t)
    xw += b_ih
    xw = xw.reshape(T, N, 3 * H)
    for t in range(T):
        hw = _matmul(h, w_hh_t) + b_hh
        r = _sigmoid(xw[t][:, 0:H] + hw[:, 0:H])
        z = _sigmoid(xw[t][:, H:2 * H] + hw[:, H:2 * H])
        n = np.tanh(xw[t][:, 2 * H:] + r * hw[:, 2 * H:])
        h = (n + z * (h - n)).astype(dtype)
        out[t] = h
    return np.ascontiguousarray(out.transpose(1, 0, 2))


def sequence_model(x, params, prefix, num_layers=2, activation=None, dtype=np.float32, cell="LSTM"):
    """SequenceModel.forward (sequence_model.py:106-125): [B, F, T] -> [B, F', T].

    ``params`` holds reference state_dict names: ``{prefix}.sequence_model.weight_ih_l{k}`` ...
    ``{prefix}.fc_output_layer.{weight,bias}``.
    """
    o = np.ascontiguousarray(np.asarray(x, dtype=dtype).transpose(0, 2, 1))  # [B, T, F]
    for k in range(num_layers):
        p = f"{prefix}.sequence_model."
        layer = lstm_layer if cell == "LSTM" else gru_layer
        o = layer(o, params[p + f"weight_ih_l{k}"], params[p + f"weight_hh_l{k}"],
                  params[p + f"bias_ih_l{k}"], params[p + f"bias_hh_l{k}"], dtype=dtype)
    w = np.asarray(params[f"{prefix}.fc_output_layer.weight"], dtype=dtype)
    b = np.asarray(params[f"{prefix}.fc_output_layer.bias"], dtype=dtype)
    o = (o @ w.T + b).astype(dtype)
    if activation == "ReLU":
        o = np.maximum(o, 0)
    elif activation == "Tanh":
        o = np.tanh(o)
    elif activation:
        raise NotImplementedError(activation)
    return np.ascontiguousarray(o.transpose(0, 2, 1))


# --------------------------------------------------------------------------- #
# FullSubNet forward      recipes/dns_interspeech_2020/fullsubnet/model.py:72-136
# --------------------------------------------------------------------------- #
def fullsubnet_forward(noisy_mag, params, look_ahead=2, sb_num_neighbors=15, fb_num_neighbors=0,
                       norm_type="offline_laplace_norm", num_groups_in_drop_band=1,
                       dtype=np.float32, return_intermediates=False, cell="LSTM", fb_activation="ReLU"):
    """noisy_mag [B, 1, F, T] -> compressed cIRM [B, 2, F, T] (or F//g under drop_band)."""
    norm = NORMS[norm_type]
    x = np.asarray(noisy_mag, dtype=dtype)
    assert x.ndim == 4 and x.shape[1] == 1
    x = np.pad(x, [(0, 0), (0, 0), (0, 0), (0, look_ahead)])  # model.py:85
    B, C, F, Tp = x.shape

    fb_input = norm(x, dtype=dtype).reshape(B, C * F, Tp)  # model.py:92-94
    fb_output = sequence_model(fb_input, params, "fb_model", activation=fb_activation, dtype=dtype, cell=cell)
    fb_output = fb_output.reshape(B, 1, F, Tp)  # model.py:95

    fb_unf = freq_unfold(fb_output, fb_num_neighbors).reshape(B, F, 2 * fb_num_neighbors + 1, Tp)
    nm_unf = freq_unfold(x, sb_num_neighbors).reshape(B, F, 2 * sb_num_neighbors + 1, Tp)
    sb_input = np.concatenate([nm_unf, fb_unf], axis=2)  # model.py:110
    sb_input = norm(sb_input, dtype=dtype)  # model.py:111

    Fs = F
    if B > 1:  # model.py:114 (runs in eval mode too, quirk Q1)
        sb_input = drop_band(sb_input.transpose(0, 2, 1, 3), num_groups_in_drop_band)
        Fs = sb_input.shape[2]
        sb_input = sb_input.transpose(0, 2, 1, 3)
    n_in = 2 * sb_num_neighbors + 1 + 2 * fb_num_neighbors + 1
    sb_in = np.ascontiguousarray(sb_input).reshape(B * Fs, n_in, Tp)  # model.py:121-125
    sb_mask = sequence_model(sb_in, params, "sb_model", activation=None, dtype=dtype, cell=cell)
    sb_mask = sb_mask.reshape(B, Fs, 2, Tp).transpose(0, 2, 1, 3)  # model.py:129-133
    out = np.ascontiguousarray(sb_mask[:, :, :, look_ahead:])  # model.py:135
    if return_intermediates:
        return out, {"fb_input": fb_input, "fb_output": fb_output, "sb_input": sb_in}
    return out


def full_band_crm_mask(noisy, params, n_fft=512, hop_length=256, win_length=512, window=None,
                       dtype=np.float32, return_intermediates=False, **model_kw):
    """recipes/dns_interspeech_2020/inferencer.py:130-145.  noisy [B, L] -> enhanced [B, L]."""
    mag, _, re, im = stft(noisy, n_fft, hop_length, win_length, window=window, dtype=dtype)
    crm = fullsubnet_forward(mag[:, None], params, dtype=dtype, **model_kw)  # [B, 2, F, T]
    crm_p = crm.transpose(0, 2, 3, 1)
    dm = decompress_cIRM(crm_p, dtype=dtype)
    er = (dm[..., 0] * re - dm[..., 1] * im).astype(dtype)
    ei = (dm[..., 1] * re + dm[..., 0] * im).astype(dtype)
    y = istft(er, ei, n_fft, hop_length, win_length, length=np.asarray(noisy).shape[-1],
              window=window, dtype=dtype)
    if return_intermediates:
        return y, {"mag": mag, "real": re, "imag": im, "crm": crm, "enh_real": er, "enh_imag": ei}
    return y


# --------------------------------------------------------------------------- #
# deterministic synthetic inputs / weights: fsn_synthetic.py (shared with bench.py and tools/, which
# may not import the oracle outside the cpu_baseline leg); re-exported here for the tests
# --------------------------------------------------------------------------- #
from fsn_synthetic import FULLSUBNET_SHAPES, make_noisy, make_params  # noqa: E402,F401
