"""Mirror of ``audio_zen/acoustics/mask.py`` on top of libfsn_hip.so."""
import torch

from .. import _lib


def _ew(fn_name, mask):
    x = mask.contiguous()
    out = torch.empty_like(x)
    fn = getattr(_lib.lib(), fn_name)
    _lib.check(fn(_lib.dev_ptr(x, "mask"), _lib.dev_ptr(out), x.numel(), _lib.stream_ptr(x.device)))
    return out


def build_complex_ideal_ratio_mask(noisy_real, noisy_imag, clean_real, clean_imag) -> torch.Tensor:
    """mask.py:7-29.  [B, F, T] x4 -> compressed cIRM [B, F, T, 2]."""
    nr, ni, cr, ci = (t.contiguous() for t in (noisy_real, noisy_imag, clean_real, clean_imag))
    out = torch.empty(nr.shape + (2,), dtype=torch.float32, device=nr.device)
    _lib.check(_lib.lib().fsn_build_cirm(_lib.dev_ptr(nr, "noisy_real"), _lib.dev_ptr(ni, "noisy_imag"),
                                         _lib.dev_ptr(cr, "clean_real"), _lib.dev_ptr(ci, "clean_imag"),
                                         _lib.dev_ptr(out), nr.numel(), _lib.stream_ptr(nr.device)))
    return out


def compress_cIRM(mask, K=10, C=0.1):
    """mask.py:32-44 (K = 10, C = 0.1 are the only values the reference uses)."""
    assert K == 10 and C == 0.1, "libfsn_hip implements the reference's K=10, C=0.1"
    return _ew("fsn_compress_cirm", mask)


def decompress_cIRM(mask, K=10, limit=9.9):
    """mask.py:47-64 (K = 10, limit = 9.9 on the path; quirk Q6)."""
    assert K == 10 and limit == 9.9, "libfsn_hip implements the reference's K=10, limit=9.9"
    return _ew("fsn_decompress_cirm", mask)


def complex_mul(noisy_r, noisy_i, mask_r, mask_i):
    """mask.py:67-70."""
    r = noisy_r * mask_r - noisy_i * mask_i
    i = noisy_r * mask_i + noisy_i * mask_r
    return r, i
