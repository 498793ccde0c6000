"""Mirror of ``audio_zen/acoustics/feature.py`` (stft :9-50, istft :53-91, mag_phase :94-96,
drop_band :309-345) on top of libfsn_hip.so.  Same names, argument meaning and return values."""
import torch

from .. import _lib

_WINDOWS = {}


def hann_window(n_fft, device):
    """``torch.hann_window(n_fft)`` exactly as the reference's CPU path evaluates it (feature.py:38):
    computed once on the CPU (ROCm's cosine may differ in the last bit) and cached per device."""
    key = (n_fft, str(device))
    w = _WINDOWS.get(key)
    if w is None:
        w = torch.hann_window(n_fft).to(device)
        _WINDOWS[key] = w
    return w


def stft(y, n_fft, hop_length, win_length, return_phase=True):
    """feature.py:9-50.  y: [B, T] or [B, C, T] -> (mag, phase, real, imag), each [B, F, T] / [B, C, F, T].
    return_phase=False (every caller inside this package: the path never reads the phase, inferencer.py:132
    discards it) skips the atan2 pass and returns None in its place."""
    num_dims = y.dim()
    assert num_dims == 2 or num_dims == 3, "Only support 2D or 3D Input"
    batch_size, num_samples = y.shape[0], y.shape[-1]
    if num_dims == 3:
        y = y.reshape(-1, num_samples)
    y = y.contiguous()
    B = y.shape[0]
    F, T = n_fft // 2 + 1, 1 + num_samples // hop_length
    real = torch.empty((B, F, T), dtype=torch.float32, device=y.device)
    imag = torch.empty_like(real)
    mag = torch.empty_like(real)
    L = _lib.lib()
    _lib.check(L.fsn_stft(_lib.dev_ptr(y, "y"), B, num_samples, n_fft, hop_length, win_length,
                          _lib.dev_ptr(hann_window(n_fft, y.device)), _lib.dev_ptr(real), _lib.dev_ptr(imag),
                          _lib.dev_ptr(mag), _lib.stream_ptr(y.device)))
    phase = torch.atan2(imag, real) if return_phase else None  # only for callers that ask: nothing on the path reads it
    if num_dims == 3:
        mag, real, imag = (t.reshape(batch_size, -1, F, T) for t in (mag, real, imag))
        phase = phase.reshape(batch_size, -1, F, T) if return_phase else None
    return mag, phase, real, imag


def istft(features, n_fft, hop_length, win_length, length=None, input_type="complex"):
    """feature.py:53-91.  features: complex [B, F, T] | (real, imag) | (mag, phase) -> [B, length]."""
    if input_type == "real_imag":
        assert isinstance(features, tuple) or isinstance(features, list)
        real, imag = features
    elif input_type == "complex":
        assert torch.is_complex(features), "The input feature is not complex."
        real, imag = features.real, features.imag
    elif input_type == "mag_phase":
        assert isinstance(features, tuple) or isinstance(features, list)
        mag, phase = features
        real, imag = mag * torch.cos(phase), mag * torch.sin(phase)
    else:
        raise NotImplementedError("Only 'real_imag', 'complex', and 'mag_phase' are supported.")
    real, imag = real.contiguous(), imag.contiguous()
    B, F, T = real.shape
    assert F == n_fft // 2 + 1
    if length is None:
        length = hop_length * (T - 1)
    y = torch.empty((B, length), dtype=torch.float32, device=real.device)
    L = _lib.lib()
    ws = _lib.workspace(L.fsn_istft_workspace_bytes(B, T, n_fft), real.device)
    _lib.check(L.fsn_istft(_lib.dev_ptr(real, "real"), _lib.dev_ptr(imag, "imag"), B, T, n_fft, hop_length,
                           win_length, _lib.dev_ptr(hann_window(n_fft, real.device)), length, _lib.dev_ptr(y),
                           ws.data_ptr(), ws.numel(), _lib.stream_ptr(real.device)))
    return y


def mag_phase(complex_tensor):
    """feature.py:94-96."""
    return torch.abs(complex_tensor), torch.angle(complex_tensor)


def drop_band(input, num_groups=2):
    """feature.py:309-345 (pure index selection).  input: [B, C, F, T] -> [B, C, F // num_groups, T]."""
    batch_size, _, num_freqs, _ = input.shape
    assert batch_size > num_groups, (
        f"Batch size = {batch_size}, num_groups = {num_groups}. The batch size should larger than the num_groups.")
    if num_groups <= 1:
        return input
    if num_freqs % num_groups != 0:
        input = input[..., : (num_freqs - (num_freqs % num_groups)), :]
        num_freqs = input.shape[2]
    output = [input[g::num_groups, :, g:num_freqs:num_groups, :] for g in range(num_groups)]
    return torch.cat(output, dim=0)
