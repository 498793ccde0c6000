"""``BaseModel`` of audio_zen/model/base_model.py for the model families composed from
``SequenceModel`` blocks (Fast FullSubNet, the full-band baseline).

Helper semantics follow the reference line by line (cited per method).  The five norms run on the HIP kernels of
norm_kernels.hip (`fsn_norm`) for GPU tensors outside autograd (inference: every composed model) and as the same
tensor algebra, autograd-tracked, inside training graphs and in the CPU unit tests; the other helpers are thin,
HBM-light glue between the LSTM blocks.  The FullSubNet model itself does not use these: its norms, unfold and
concat are fused into the HIP kernels (fullsubnet_amd/model.py).
"""
import torch
import torch.nn as nn
from torch.nn import functional

from .acoustics.feature import drop_band

EPSILON = float(torch.finfo(torch.float32).eps)  # audio_zen/constant.py


def _hip_norm(name, input, sample_length=192, eps=0.0):
    """fsn_norm (norm_kernels.hip) when the tensor lives on the GPU and nothing asks for a gradient; None otherwise (the
    caller then runs the same norm as autograd-tracked tensor algebra: training graphs, and the CPU unit tests).
    eps = 0: the constant of audio_zen/model/base_model.py for that norm."""
    from . import _lib
    if not input.is_cuda or (torch.is_grad_enabled() and input.requires_grad) or input.dtype != torch.float32:
        return None
    if input.shape[-1] > 6144:  # FSN_NORM_MAX_FRAMES: longer inputs take the tensor algebra below
        return None
    offline = name in ("offline_laplace_norm", "offline_gaussian_norm")
    if input.dim() != 4 and not (offline and input.dim() >= 2):
        return None
    x = input.contiguous()
    if input.dim() == 4:
        B, C, F, T = x.shape
    else:  # the offline norms take their statistic over everything but the batch axis, whatever the rank
        B, C, T = x.shape[0], 1, x.shape[-1]
        F = x.numel() // (B * T)
    L = _lib.lib()
    nt = _lib.ALL_NORM_TYPES[name]
    y = torch.empty_like(x)
    ws = _lib.workspace(L.fsn_norm_workspace_bytes(nt, B, C, F, T), x.device)
    _lib.check(L.fsn_norm(_lib.dev_ptr(x, "input"), _lib.dev_ptr(y), nt, B, C, F, T, int(sample_length), float(eps),
                          ws.data_ptr(), ws.numel(), _lib.stream_ptr(x.device)))
    return y


class BaseModel(nn.Module):
    def __init__(self):
        super().__init__()

    # base_model.py:14-46
    @staticmethod
    def freq_unfold(input, num_neighbors):
        """[B, C, F, T] -> [B, F, C, 2n+1, T]: unit f sees bins reflect(f - n .. f + n)."""
        assert input.dim() == 4, f"The dim of the input is {input.dim()}. It should be four dim."
        B, C, F, T = input.shape
        if num_neighbors <= 0:
            return input.permute(0, 2, 1, 3).reshape(B, F, C, 1, T)
        n = num_neighbors
        idx = torch.arange(F, device=input.device).reshape(F, 1) + torch.arange(-n, n + 1, device=input.device)
        idx = idx.abs()
        idx = torch.where(idx > F - 1, 2 * (F - 1) - idx, idx)  # reflect without repeating the edge
        out = input[:, :, idx, :]  # [B, C, F, 2n+1, T]
        return out.permute(0, 2, 1, 3, 4).contiguous()

    # base_model.py:204-218
    @staticmethod
    def offline_laplace_norm(input):
        y = _hip_norm("offline_laplace_norm", input)
        if y is not None:
            return y
        mu = torch.mean(input, dim=list(range(1, input.dim())), keepdim=True)
        return input / (mu + 1e-5)

    # base_model.py:221-251
    @staticmethod
    def cumulative_laplace_norm(input):
        y = _hip_norm("cumulative_laplace_norm", input)
        if y is not None:
            return y
        B, C, F, T = input.size()
        x = input.reshape(B * C, F, T)
        cum = torch.cumsum(torch.sum(x, dim=1), dim=-1)
        count = torch.arange(F, F * T + 1, F, dtype=x.dtype, device=x.device).reshape(1, T)
        mean = (cum / count).reshape(B * C, 1, T)
        return (x / (mean + EPSILON)).reshape(B, C, F, T)

    # base_model.py:295-310
    @staticmethod
    def offline_gaussian_norm(input):
        y = _hip_norm("offline_gaussian_norm", input)
        if y is not None:
            return y
        mu = torch.mean(input, dim=(1, 2, 3), keepdim=True)
        std = torch.std(input, dim=(1, 2, 3), keepdim=True)
        return (input - mu) / (std + 1e-5)

    # base_model.py:312-354
    @staticmethod
    def cumulative_layer_norm(input):
        y = _hip_norm("cumulative_layer_norm", input)
        if y is not None:
            return y
        B, C, F, T = input.size()
        x = input.reshape(B * C, F, T)
        s1 = torch.cumsum(torch.sum(x, dim=1), dim=-1)
        s2 = torch.cumsum(torch.sum(torch.square(x), dim=1), dim=-1)
        count = torch.arange(F, F * T + 1, F, dtype=x.dtype, device=x.device).reshape(1, T)
        mean = s1 / count
        var = (s2 - 2 * mean * s1) / count + mean.pow(2)
        std = torch.sqrt(var + EPSILON)
        return ((x - mean.reshape(B * C, 1, T)) / std.reshape(B * C, 1, T)).reshape(B, C, F, T)

    # base_model.py:103-151
    @staticmethod
    def forgetting_norm(input, sample_length=192):
        """mu_t = a_t mu_{t-1} + (1 - a_t) mean_f x[:, :, t] with a_t = min((t-1)/(t+1), alpha) for
        t < sample_length (a_0 = -1, i.e. mu_0 = 2 mean_0 - the reference's own start-up) and alpha after."""
        assert input.ndim == 4
        y = _hip_norm("forgetting_norm", input, sample_length)
        if y is not None:
            return y
        B, C, F, T = input.size()
        x = input.reshape(B, C * F, T)
        frame_mean = torch.mean(x, dim=1)  # [B, T]
        alpha = (sample_length - 1) / (sample_length + 1)
        mu = torch.zeros((B,), dtype=x.dtype, device=x.device)
        mus = []
        for t in range(T):
            # start-up factor formed in fp32 like the reference's torch.min(torch.tensor([..., alpha]))
            a_t = torch.tensor([(t - 1) / (t + 1), alpha]).min() if t < sample_length else alpha
            mu = a_t * mu + (1 - a_t) * frame_mean[:, t]
            mus.append(mu)
        mu = torch.stack(mus, dim=-1).reshape(B, 1, T)
        return (x / (mu + 1e-10)).reshape(B, C, F, T)

    # base_model.py:356-372
    def norm_wrapper(self, norm_type: str):
        norms = {
            "offline_laplace_norm": self.offline_laplace_norm,
            "cumulative_laplace_norm": self.cumulative_laplace_norm,
            "offline_gaussian_norm": self.offline_gaussian_norm,
            "cumulative_layer_norm": self.cumulative_layer_norm,
            "forgetting_norm": self.forgetting_norm,
        }
        if norm_type not in norms:
            raise NotImplementedError("You must set up a type of Norm. "
                                      "e.g. offline_laplace_norm, cumulative_laplace_norm, forgetting_norm, etc.")
        return norms[norm_type]

    # base_model.py:374-439
    def weight_init(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight.data)
            nn.init.normal_(m.bias.data)
        elif isinstance(m, (nn.LSTM, nn.GRU)):
            for param in m.parameters():
                if len(param.shape) >= 2:
                    nn.init.orthogonal_(param.data)
                else:
                    nn.init.normal_(param.data)

    # base_model.py:254-292 (duplicate of feature.py:309-345)
    @staticmethod
    def drop_band(input, num_groups=2):
        return drop_band(input, num_groups)


def look_ahead_pad(x, look_ahead):
    """functional.pad(x, [0, look_ahead]) (fullsubnet/model.py:85 and every sibling model)."""
    return functional.pad(x, [0, look_ahead])
