"""Two stacked GRU layers with few rows (the full-band block of a GRU FullSubNet: 257 -> 512 x 2 -> 257): ONE persistent launch of
the chain kernel (fsn_gru2_forward) against the layer-by-layer path (gru_step_kernel, 2 T launches).  usage: bench_gru_chain.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402,F401
from fullsubnet_amd import sequence_model as SM  # noqa: E402

m = SM.SequenceModel(257, 257, 512, 2, False, "GRU", "ReLU").cuda().eval()
L = SM._lib.lib()
supported = L.fsn_gru2_forward_supported


def timed(x):
    for _ in range(3):
        y = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = m(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3, y


with torch.no_grad():
    for B in (1, 8, 64):
        x = torch.randn(B, 257, 190).cuda()
        a, ya = timed(x)
        L.fsn_gru2_forward_supported = lambda *args: 0  # the rows "do not fit": layer by layer
        b, yb = timed(x)
        L.fsn_gru2_forward_supported = supported
        print(f"{B} utterance(s) x 190 frames: chain {a:.2f} ms, layer by layer {b:.2f} ms, max |d| {(ya - yb).abs().max().item():.1e}")
