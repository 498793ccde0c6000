"""hipGraph replay of the whole enhancement path for small batches (the launch-chain regime):
capture Model.enhance once with torch.cuda.CUDAGraph (the library only enqueues on the caller's stream and forks /
joins its auxiliary stream with events, so the call is capturable), replay it, compare with the eager call.
usage: python tools/bench_graph.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model  # noqa: E402
from fsn_synthetic import make_noisy  # noqa: E402

dev = torch.device("cuda", 0)
model, _ = build_model(dev)
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 8]:
    noisy = torch.from_numpy(make_noisy(B, 48000, seed=7)).to(dev)
    for _ in range(3):
        eager = model.enhance(noisy)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        eager = model.enhance(noisy)
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / K
    static_in = noisy.clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        model.enhance(static_in)  # warm-up on the capture stream
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.cuda.graph(g):
        static_out = model.enhance(static_in)
    static_in.copy_(noisy)
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(static_out, eager)
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / K
    print(f"B={B}: eager {t_eager * 1e3:.2f} ms, graph replay {t_graph * 1e3:.2f} ms, bit-identical={same}")
