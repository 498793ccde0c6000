"""Stage-by-stage comparison of the HIP path against the oracle (debug aid, run on the GPU box).
Re-derives the workspace carving of fsn_fullsubnet_forward to look at intermediate buffers."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd as fsn  # noqa: E402
from fullsubnet_amd import _lib  # noqa: E402
from oracle import fullsubnet_oracle as O  # noqa: E402


def rup(x, m):
    return (x + m - 1) // m * m


def main(B=2, Ls=2048, norm="offline_laplace_norm"):
    params = O.make_params(seed=0, gain=2.0, mask_gain=24.0)
    model = fsn.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                      fb_output_activate_function="ReLU", sb_output_activate_function=False,
                      fb_model_hidden_size=512, sb_model_hidden_size=384, norm_type=norm,
                      num_groups_in_drop_band=1, weight_init=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.cuda().eval()
    noisy = O.make_noisy(B, Ls, seed=1)
    win = torch.hann_window(512).numpy()
    mag, _, re, im = O.stft(noisy, window=win)
    gmag, _, gre, gim = fsn.stft(torch.from_numpy(noisy).cuda(), 512, 256, 512)
    print("stft  max|d re|", np.abs(gre.cpu().numpy() - re).max(), " |re|max", np.abs(re).max())
    print("stft  max|d mag|", np.abs(gmag.cpu().numpy() - mag).max())

    crm_ref, inter = O.fullsubnet_forward(mag[:, None], params, norm_type=norm, return_intermediates=True)
    F, FP, la, Hf, Hs = 257, 272, 2, 512, 384
    T = mag.shape[-1]
    Tp = T + la
    L = _lib.lib()
    cfg = model._cfg
    x = torch.from_numpy(mag[:, None].copy()).cuda()
    out = torch.empty((B, 2, F, T), device="cuda")
    nbytes = L.fsn_fullsubnet_workspace_bytes(ctypes.byref(cfg), B, T)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    _lib.check(L.fsn_fullsubnet_forward(ctypes.byref(cfg), model.packed_weights().data_ptr(), x.data_ptr(), B, T,
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    off = [0]

    def take(count, dtype=torch.float32):
        off[0] = rup(off[0], 256)
        nb = count * (8 if dtype == torch.float64 else 4)
        t = ws[off[0]: off[0] + nb].view(dtype)
        off[0] += nb
        return t

    Npad_fb = rup(B, 16)
    N = B * F
    tiles = (N + 15) // 16
    best = None
    for rt in range(1, 6):
        wgs = (tiles + rt - 1) // rt
        rounds = (wgs + 255) // 256
        cost = rounds * rt * 64 + rounds
        if best is None or cost < best[0] or (cost == best[0] and rt > best[1]):
            best = (cost, rt)
    RT = best[1]
    Npad = rup(N, 16 * RT)
    print("RT", RT, "Npad", Npad)
    magT = take(B * Tp * FP).view(B, Tp, FP)
    crm_r = take(B * T * FP).view(B, T, FP)
    crm_i = take(B * T * FP).view(B, T, FP)
    gx_fb = take(Tp * Npad_fb * 4 * Hf)
    hseq_fb0 = take(Tp * Npad_fb * Hf).view(Tp, Npad_fb, Hf)
    hseq_fb1 = take(Tp * Npad_fb * Hf).view(Tp, Npad_fb, Hf)
    c_fb = take(Npad_fb * Hf)
    fb_out = take(B * Tp * FP).view(B, Tp, FP)
    binsum = take(B * FP, torch.float64).view(B, FP)
    cum = norm != "offline_laplace_norm"
    den_fb = take(B * Tp if cum else B)
    den_sb = take(Tp * Npad if cum else B)
    gx_sb = take(Tp * Npad * 4 * Hs)
    hseq_sb0 = take(Tp * Npad * Hs).view(Tp, Npad, Hs)
    hseq_sb1 = take(Tp * Npad * Hs).view(Tp, Npad, Hs)

    magp = np.pad(mag, [(0, 0), (0, 0), (0, la)])
    print("magT  max|d|", np.abs(magT[:, :, :F].cpu().numpy() - magp.transpose(0, 2, 1)).max(),
          "pad", magT[:, :, F:].abs().max().item())
    if not cum:
        mu = magp.mean(axis=(1, 2))
        print("den_fb", den_fb.cpu().numpy(), "ref", mu + 1e-5)
    # full-band LSTM layer 0 hidden sequence
    p = "fb_model.sequence_model."
    fb_in = inter["fb_input"].transpose(0, 2, 1)  # [B, Tp, F]
    h0 = O.lstm_layer(fb_in, params[p + "weight_ih_l0"], params[p + "weight_hh_l0"], params[p + "bias_ih_l0"],
                      params[p + "bias_hh_l0"])
    g0 = hseq_fb0[:, :B].cpu().numpy().transpose(1, 0, 2)
    print("fb h0 max|d|", np.abs(g0 - h0).max(), " step0", np.abs(g0[:, 0] - h0[:, 0]).max(), " step1",
          np.abs(g0[:, 1] - h0[:, 1]).max())
    h1 = O.lstm_layer(h0, params[p + "weight_ih_l1"], params[p + "weight_hh_l1"], params[p + "bias_ih_l1"],
                      params[p + "bias_hh_l1"])
    g1 = hseq_fb1[:, :B].cpu().numpy().transpose(1, 0, 2)
    print("fb h1 max|d|", np.abs(g1 - h1).max())
    fbo = inter["fb_output"][:, 0].transpose(0, 2, 1)
    print("fb_out max|d|", np.abs(fb_out[:, :, :F].cpu().numpy() - fbo).max(), "ref max", np.abs(fbo).max())
    if not cum:
        print("den_sb", den_sb.cpu().numpy())
    sb_in = inter["sb_input"].transpose(0, 2, 1)  # [N, Tp, 32]
    p = "sb_model.sequence_model."
    s0 = O.lstm_layer(sb_in, params[p + "weight_ih_l0"], params[p + "weight_hh_l0"], params[p + "bias_ih_l0"],
                      params[p + "bias_hh_l0"])
    gs0 = hseq_sb0[:, :N].cpu().numpy().transpose(1, 0, 2)
    print("sb h0 max|d|", np.abs(gs0 - s0).max(), " step0", np.abs(gs0[:, 0] - s0[:, 0]).max(), " step1",
          np.abs(gs0[:, 1] - s0[:, 1]).max())
    s1 = O.lstm_layer(s0, params[p + "weight_ih_l1"], params[p + "weight_hh_l1"], params[p + "bias_ih_l1"],
                      params[p + "bias_hh_l1"])
    gs1 = hseq_sb1[:, :N].cpu().numpy().transpose(1, 0, 2)
    print("sb h1 max|d|", np.abs(gs1 - s1).max())
    print("crm   max|d|", np.abs(out.cpu().numpy() - crm_ref).max(), " ref range", crm_ref.min(), crm_ref.max())
    err = np.abs(out.cpu().numpy() - crm_ref)
    print("crm worst idx", np.unravel_index(err.argmax(), err.shape))


if __name__ == "__main__":
    main(norm=sys.argv[1] if len(sys.argv) > 1 else "offline_laplace_norm")
