"""CPU oracle for the TRAINING step of the FullSubNet recipe (test infrastructure only, see the
header of fullsubnet_oracle.py for the rules).

Restates recipes/dns_interspeech_2020/fullsubnet/trainer.py:41-71 with use_amp = false:
    stft(noisy), stft(clean) -> compressed cIRM target -> drop_band -> Model.forward (drop_band
    inside, fullsubnet/model.py:72-136) -> MSELoss -> backward -> clip_grad_norm_(10) -> Adam.
The nn.LSTM layers are restated as an explicit loop over time on torch CPU tensors (gate order
i, f, g, o; sequence_model.py:52-58) so that autograd differentiates the *restated* cell, not ATen's
fused LSTM; everything else uses the same tensor algebra as the reference.  Pinned against the
reference's own loss / gradients / updated parameters in tests/golden/fsn_train_b4.npz and, for the cumulative Laplace
norm of train_cumulativeLaplaceNorm.toml, fsn_train_cum_b4.npz (tests/test_train_oracle.py).
"""
import numpy as np
import torch
import torch.nn.functional as functional

EPSILON = float(np.finfo(np.float32).eps)


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh):
    """x [N, T, I] (batch_first) -> [N, T, H]; explicit cell, h0 = c0 = 0."""
    N, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros((N, H))
    c = x.new_zeros((N, H))
    xw = x @ w_ih.t() + (b_ih + b_hh)
    outs = []
    for t in range(T):
        gates = xw[:, t] + h @ w_hh.t()
        i, f, g, o = gates.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


def sequence_model(x, p, prefix, relu):
    """sequence_model.py:106-125: [B, F, T] -> [B, F', T]."""
    o = x.permute(0, 2, 1)
    for k in range(2):
        q = f"{prefix}.sequence_model."
        o = lstm_layer(o, p[q + f"weight_ih_l{k}"], p[q + f"weight_hh_l{k}"], p[q + f"bias_ih_l{k}"],
                       p[q + f"bias_hh_l{k}"])
    o = o @ p[f"{prefix}.fc_output_layer.weight"].t() + p[f"{prefix}.fc_output_layer.bias"]
    if relu:
        o = torch.relu(o)
    return o.permute(0, 2, 1)


def freq_unfold(x, n):
    """base_model.py:14-46."""
    B, C, F, T = x.shape
    if n <= 0:
        return x.permute(0, 2, 1, 3).reshape(B, F, C, 1, T)
    out = functional.pad(x.reshape(B * C, 1, F, T), [0, 0, n, n], mode="reflect")
    out = functional.unfold(out, kernel_size=(2 * n + 1, T))
    return out.reshape(B, C, 2 * n + 1, T, F).permute(0, 4, 1, 2, 3).contiguous()


def offline_laplace_norm(x):
    mu = torch.mean(x, dim=list(range(1, x.dim())), keepdim=True)
    return x / (mu + 1e-5)


def cumulative_laplace_norm(x):
    """base_model.py:221-251: x [B, C, F, T] -> x / (running mean over (F, frames <= t) of every (b, c) + EPSILON).  On the
    4-D sub-band tensor [B, F, 2 nb + 2, T'] the units f take the place of the channels."""
    B, C, F, T = x.shape
    xr = x.reshape(B * C, F, T)
    cum = torch.cumsum(torch.sum(xr, dim=1), dim=-1)
    count = torch.arange(F, F * T + 1, F, dtype=x.dtype).reshape(1, T)
    mean = (cum / count).reshape(B * C, 1, T)
    return (xr / (mean + EPSILON)).reshape(B, C, F, T)


NORMS = {"offline_laplace_norm": offline_laplace_norm, "cumulative_laplace_norm": cumulative_laplace_norm}


def drop_band(x, g):
    """feature.py:309-345."""
    B, _, F, _ = x.shape
    if g <= 1:
        return x
    if F % g:
        x = x[..., : F - F % g, :]
        F = x.shape[2]
    return torch.cat([x[i::g][:, :, i:F:g, :] for i in range(g)], dim=0)


def model_forward(noisy_mag, p, look_ahead=2, nb=15, groups=2, norm_type="offline_laplace_norm"):
    """fullsubnet/model.py:72-136."""
    norm = NORMS[norm_type]
    x = functional.pad(noisy_mag, [0, look_ahead])
    B, C, F, Tp = x.shape
    fb_out = sequence_model(norm(x).reshape(B, F, Tp), p, "fb_model", True).reshape(B, 1, F, Tp)
    sb_in = torch.cat([freq_unfold(x, nb).reshape(B, F, 2 * nb + 1, Tp), freq_unfold(fb_out, 0).reshape(B, F, 1, Tp)],
                      dim=2)
    sb_in = norm(sb_in)
    Fs = F
    if B > 1:
        sb_in = drop_band(sb_in.permute(0, 2, 1, 3), groups)
        Fs = sb_in.shape[2]
        sb_in = sb_in.permute(0, 2, 1, 3)
    m = sequence_model(sb_in.reshape(B * Fs, 2 * nb + 2, Tp), p, "sb_model", False)
    m = m.reshape(B, Fs, 2, Tp).permute(0, 2, 1, 3).contiguous()
    return m[:, :, :, look_ahead:]


def compress_cirm(m):
    m = -100 * (m <= -100) + m * (m > -100)
    return 10 * (1 - torch.exp(-0.1 * m)) / (1 + torch.exp(-0.1 * m))


def train_step(params, noisy, clean, groups=2, lr=1e-3, clip=10.0, norm_type="offline_laplace_norm"):
    """One iteration; returns dict(loss, grads {name: tensor}, new_params {name: tensor}).
    params: {reference state_dict name: np.ndarray}; noisy / clean: np.ndarray [B, L]."""
    p = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in params.items()}
    win = torch.hann_window(512)
    sn = torch.stft(torch.from_numpy(noisy), 512, 256, 512, window=win, return_complex=True)
    sc = torch.stft(torch.from_numpy(clean), 512, 256, 512, window=win, return_complex=True)
    den = sn.real ** 2 + sn.imag ** 2 + EPSILON
    cirm = compress_cirm(torch.stack(((sn.real * sc.real + sn.imag * sc.imag) / den,
                                      (sn.real * sc.imag - sn.imag * sc.real) / den), dim=-1))
    cirm = drop_band(cirm.permute(0, 3, 1, 2), groups).permute(0, 2, 3, 1)
    crm = model_forward(sn.abs().unsqueeze(1), p, groups=groups, norm_type=norm_type).permute(0, 2, 3, 1)
    loss = torch.mean((cirm - crm) ** 2)
    loss.backward()
    names = list(p)
    torch.nn.utils.clip_grad_norm_([p[k] for k in names], clip)
    grads = {k: p[k].grad.detach().clone() for k in names}
    opt = torch.optim.Adam([p[k] for k in names], lr=lr, betas=(0.9, 0.999))
    opt.step()
    return dict(loss=float(loss.detach()), grads=grads, new_params={k: p[k].detach().clone() for k in names})
