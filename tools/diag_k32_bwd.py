"""Diagnosis of the 16-bit BPTT kernel: every gradient of tests/test_gpu_amp.py::test_two_layer_lstm_16bit_operands_vs_an_exact_emulation
against the fp32 mode, twice (determinism).  usage: diag_k32_bwd.py [T] [--lib tools/bin/<variant>.so] (tools/build_variant.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    fullsubnet_amd._lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from fullsubnet_amd.train import Lstm2Function  # noqa: E402

T, N, I, H = int(sys.argv[1]) if len(sys.argv) > 1 else 5, 2048, 32, 384
g = torch.Generator().manual_seed(11)
k = 1.0 / np.sqrt(H)
x = torch.randn(T, N, I, generator=g)
shapes = ((4 * H, I), (4 * H, H), (4 * H,), (4 * H,), (4 * H, H), (4 * H, H), (4 * H,), (4 * H,))
w = [(torch.rand(s_, generator=g) * 2 - 1) * k * 2 for s_ in shapes]
dy = torch.randn(T, N, H, generator=g) * 64.0
names = ["y", "dx", "dw_ih0", "dw_hh0", "db_ih0", "db_hh0", "dw_ih1", "dw_hh1", "db_ih1", "db_hh1"]


def run(arith):
    xd = x.cuda().requires_grad_(True)
    wd = [t.cuda().requires_grad_(True) for t in w]
    y = Lstm2Function.apply(xd, *wd, arith)
    (y * dy.cuda()).sum().backward()
    torch.cuda.synchronize()
    return [y.detach().cpu()] + [xd.grad.cpu()] + [t.grad.cpu() for t in wd]


ref = run("f32")
for arith in ("f16", "bf16"):
    a, b = run(arith), run(arith)
    for name, u, v, r in zip(names, a, b, ref):
        scale = max(r.abs().max().item(), 1e-3)
        err = (u - r).abs().max().item() / scale
        # where along t / rows is the deviation?
        extra = ""
        if name == "dx":
            d = (u - r).abs()
            bad = (d.amax(dim=(0, 2)) > 1e-2 * scale).nonzero().flatten()
            extra = (f" per step {[round(d[t].max().item() / scale, 4) for t in range(T)]} rows with > 1e-2: {len(bad)}; row % 64 histogram "
                     f"{torch.bincount(bad % 64, minlength=64).tolist()}; clusters hit {len(set((bad // 64).tolist()))}; "
                     f"twice-different rows {int(((u - v).abs().amax(dim=(0, 2)) > 0).sum())}")
        print(f"{arith} {name:7s}: vs fp32 {err:.2e}  same twice: {torch.equal(u, v)}{extra}")
