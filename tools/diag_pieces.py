"""Experiment: a two-layer H = 512 LSTM stack with FEW rows (Fast FullSubNet's decoder pair in training: 72 rows) as pieces of 16
rows on concurrent streams through the chain kernels (fsn_lstm2_forward_train / fsn_lstm2_backward: fb_chain_kernel +
fb_chain_bptt_kernel, 16 rows each) against the layer-by-layer per-step launches.  usage: diag_pieces.py [rows] [I] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fullsubnet_amd.train import Lstm2Function, LstmLayerFunction  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 72
I = int(sys.argv[2]) if len(sys.argv) > 2 else 128
T = int(sys.argv[3]) if len(sys.argv) > 3 else 195
H = 512
torch.manual_seed(0)
lstm = torch.nn.LSTM(I, H, num_layers=2).cuda()
x = torch.randn(T, N, I, device="cuda", requires_grad=True)
g = torch.randn(T, N, H, device="cuda")
params = [getattr(lstm, f"{n}_l{k}") for k in (0, 1) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


def layerwise():
    h = x
    for k in (0, 1):
        h = LstmLayerFunction.apply(h, *params[4 * k:4 * k + 4])
    return h


streams = [torch.cuda.Stream() for _ in range(8)]


def pieces(concurrent=True):
    n = (N + 15) // 16
    xp = torch.nn.functional.pad(x, (0, 0, 0, 16 * n - N))
    cur = torch.cuda.current_stream()
    outs = []
    for k in range(n):
        st = streams[k % len(streams)] if concurrent else cur
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(Lstm2Function.apply(xp[:, 16 * k:16 * (k + 1)], *params, "f32"))
    for k in range(n):
        cur.wait_stream(streams[k % len(streams)])
    return torch.cat(outs, dim=1)[:, :N]


def run(fn, reps=5):
    res = None
    for it in range(reps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        for p in params:
            p.grad = None
        x.grad = None
        h = fn()
        torch.cuda.synchronize() if it < 0 else None
        (h * g).sum().backward()
        res = (h.detach().clone(), x.grad.detach().clone(), [p.grad.detach().clone() for p in params])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, res


t_a, a = run(layerwise)
t_b, b = run(lambda: pieces(True))
t_c, c = run(lambda: pieces(False))
t_d, dd = run(lambda: Lstm2Function.apply(x, *params, "f32"))
d = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
print(f"rows {N}, I {I}, T {T}: layer by layer {t_a:.2f} ms, 16-row pieces on concurrent streams {t_b:.2f} ms, pieces in line {t_c:.2f} ms")
print(f"  Lstm2Function on all rows (the BPTT chain walks {(N + 15) // 16} row tiles in one launch): {t_d:.2f} ms; vs layer by layer: h {d(dd[0], a[0]):.1e}, "
      f"dx {d(dd[1], a[1]):.1e}, worst dW {max(d(u, v) for u, v in zip(dd[2], a[2])):.1e}")
print(f"  pieces vs layer by layer: h {d(b[0], a[0]):.1e}, dx {d(b[1], a[1]):.1e}, worst dW {max(d(u, v) for u, v in zip(b[2], a[2])):.1e}")
