"""Experiment: Fast FullSubNet's training step as MICRO-BATCHES in flight on concurrent streams (the model has no cross-utterance
term; its encoder / decoder blocks are chains of small per-step launches that leave most of the chip idle).
usage: diag_micro.py [batch] [micro-batches] [f16|f32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
import bench_family_train as BFT  # noqa: E402
from fullsubnet_amd.acoustics.feature import stft  # noqa: E402
from fullsubnet_amd.acoustics.mask import build_complex_ideal_ratio_mask  # noqa: E402
from fullsubnet_amd.train import mse_loss  # noqa: E402
from fsn_synthetic import make_noisy  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 72
M = int(sys.argv[2]) if len(sys.argv) > 2 else 3
arith = sys.argv[3] if len(sys.argv) > 3 else "f16"
model, _ = BFT.build("fast")
model.train_arithmetic = arith
params = [p for p in model.parameters() if p.requires_grad]
noisy = torch.from_numpy(make_noisy(B, 49152, seed=1)).cuda()
clean = torch.from_numpy(0.7 * make_noisy(B, 49152, seed=2)).cuda()
mag, _, nr, ni = stft(noisy, 512, 256, 512, return_phase=False)
_, _, cr, ci = stft(clean, 512, 256, 512, return_phase=False)
cirm = build_complex_ideal_ratio_mask(nr, ni, cr, ci)
scale = 1024.0
streams = [torch.cuda.Stream() for _ in range(M)]


def whole():
    crm = model(mag.unsqueeze(1)).permute(0, 2, 3, 1)
    return torch.autograd.grad(mse_loss(crm, cirm) * scale, params)


def micro():
    cur = torch.cuda.current_stream()
    n = B // M
    outs = []
    for k in range(M):
        st = streams[k]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            crm = model(mag[k * n:(k + 1) * n].unsqueeze(1)).permute(0, 2, 3, 1)
            loss = mse_loss(crm, cirm[k * n:(k + 1) * n]) * (scale / M)
            outs.append(torch.autograd.grad(loss, params))
    for st in streams:
        cur.wait_stream(st)
    return [sum(g) for g in zip(*outs)]


def run(fn, reps=3):
    for it in range(reps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        g = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, g


ta, ga = run(whole)
tb, gb = run(micro)
num = sum(float(((a - b) ** 2).sum()) for a, b in zip(ga, gb)) ** 0.5
den = sum(float((a ** 2).sum()) for a in ga) ** 0.5
print(f"fast B={B} {arith}: forward + backward of the whole batch {ta:.1f} ms; as {M} micro-batches on concurrent streams {tb:.1f} ms; "
      f"gradient difference {num / den:.1e} of the gradient's norm")
