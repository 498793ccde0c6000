"""ATen restatement of the reference path for the `cpu_baseline` leg of bench.py.

TEST INFRASTRUCTURE ONLY (same rule as fullsubnet_oracle.py: imported by tests/ and by bench.py's
cpu_baseline leg, never by anything under fullsubnet_amd/).

The reference checkout does not exist on the GPU box, so its CPU speed cannot be measured there directly.  What
the reference executes on this path is, however, a short list of ATen operators - and those ship with the
PyTorch build that is on the GPU box.  This module calls exactly that operator sequence, so timing it on the GPU
box's host cores is timing the reference's own CPU arithmetic (oneDNN `mkldnn_rnn_layer` for nn.LSTM, MKL for the
FFTs), not a slower re-implementation:

  torch.stft(y, 512, 256, 512, hann, return_complex=True), abs           audio_zen/acoustics/feature.py:33-49
  F.pad look-ahead, offline Laplace norm  x / (mean + 1e-5)               fullsubnet/model.py:85-92, base_model.py:204-218
  nn.LSTM(257, 512, 2, batch_first) + nn.Linear(512, 257) + ReLU          sequence_model.py:52-58,82-84,116-123
  F.pad reflect + F.unfold(kernel (31, T')) -> [B, F, 31, T'], cat, norm   base_model.py:31-44, model.py:98-111
  nn.LSTM(32, 384, 2, batch_first) + nn.Linear(384, 2) on B F sequences   model.py:121-128
  decompress_cIRM, complex mask, torch.istft                               mask.py:47-64, inferencer.py:137-143

It is pinned like the numpy oracle: tests/test_oracle_golden.py holds it to the golden vectors produced by the
reference itself (it reproduces them to rounding, being the same ATen kernels).  Inference semantics of BASELINE
config 2: every utterance keeps its full 257-bin mask (num_groups_in_drop_band = 1).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class AtenFullSubNet(nn.Module):
    def __init__(self, params, num_freqs=257, look_ahead=2, sb_num_neighbors=15, fb_hidden=512, sb_hidden=384):
        super().__init__()
        self.F, self.la, self.nb = num_freqs, look_ahead, sb_num_neighbors
        self.fb_lstm = nn.LSTM(num_freqs, fb_hidden, 2, batch_first=True)
        self.fb_fc = nn.Linear(fb_hidden, num_freqs)
        self.sb_lstm = nn.LSTM(2 * sb_num_neighbors + 2, sb_hidden, 2, batch_first=True)
        self.sb_fc = nn.Linear(sb_hidden, 2)
        sd = {}
        for k, v in params.items():
            head, rest = k.split(".", 1)
            name = rest.replace("sequence_model.", "").replace("fc_output_layer.", "")
            mod = ("fb_" if head == "fb_model" else "sb_") + ("lstm" if "sequence_model" in rest else "fc")
            sd[f"{mod}.{name}"] = torch.as_tensor(v)
        self.load_state_dict(sd, strict=True)

    @staticmethod
    def _norm(x):  # base_model.py:204-218
        mu = torch.mean(x, dim=list(range(1, x.dim())), keepdim=True)
        return x / (mu + 1e-5)

    def forward(self, mag):
        """mag [B, F, T] -> compressed cIRM [B, 2, F, T] (model.py:72-136, no band dropping)."""
        B, Fq, T = mag.shape
        x = F.pad(mag.unsqueeze(1), [0, self.la])            # [B, 1, F, T']
        Tp = T + self.la
        fb_in = self._norm(x).reshape(B, Fq, Tp)
        self.fb_lstm.flatten_parameters()
        o, _ = self.fb_lstm(fb_in.permute(0, 2, 1))
        fb_out = torch.relu(self.fb_fc(o)).permute(0, 2, 1).reshape(B, 1, Fq, Tp)
        n = self.nb
        xp = F.pad(x.reshape(B, 1, Fq, Tp), [0, 0, n, n], mode="reflect")
        unf = F.unfold(xp, kernel_size=(2 * n + 1, Tp)).reshape(B, 1, 2 * n + 1, Tp, Fq).permute(0, 4, 1, 2, 3)
        sb_in = torch.cat([unf.reshape(B, Fq, 2 * n + 1, Tp), fb_out.reshape(B, Fq, 1, Tp)], dim=2)
        sb_in = self._norm(sb_in).reshape(B * Fq, 2 * n + 2, Tp)
        self.sb_lstm.flatten_parameters()
        o, _ = self.sb_lstm(sb_in.permute(0, 2, 1))
        m = self.sb_fc(o).permute(0, 2, 1)                    # [B F, 2, T']
        m = m.reshape(B, Fq, 2, Tp).permute(0, 2, 1, 3).contiguous()
        return m[:, :, :, self.la:]


def decompress_cirm(mask, K=10.0, limit=9.9):  # mask.py:47-64
    mask = limit * (mask >= limit) - limit * (mask <= -limit) + mask * (torch.abs(mask) < limit)
    return -K * torch.log((K - mask) / (K + mask))


@torch.no_grad()
def full_band_crm_mask(model, noisy, n_fft=512, hop=256, return_crm=False):
    """inferencer.py:130-145.  noisy [B, L] (CPU tensor) -> enhanced [B, L]."""
    win = torch.hann_window(n_fft)
    spec = torch.stft(noisy, n_fft, hop, n_fft, window=win, return_complex=True)
    crm = model(torch.abs(spec))
    m = decompress_cirm(crm.permute(0, 2, 3, 1))
    er = m[..., 0] * spec.real - m[..., 1] * spec.imag
    ei = m[..., 1] * spec.real + m[..., 0] * spec.imag
    y = torch.istft(torch.complex(er, ei), n_fft, hop, n_fft, window=win, length=noisy.size(-1))
    return (y, crm) if return_crm else y
