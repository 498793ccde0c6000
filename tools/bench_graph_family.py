"""hipGraph replay of a sibling model's whole enhancement call at small batches (the launch-bound regime: ~130 kernels of
4 - 50 us around a few persistent launches): capture tools/bench_family.py's `enhance` once with torch.cuda.CUDAGraph -
the library only enqueues on the caller's streams and forks / joins with events, so the call is capturable, side streams
included - replay it, compare with the eager call.
usage: python tools/bench_graph_family.py [improved48|improved16|fast] [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_family import build, collector_paused, enhance_fn  # noqa: E402
from fsn_synthetic import make_noisy  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "improved48"
dev = torch.device("cuda", 0)
model, _, L, hop, sr, la = build(which, dev)
enhance = enhance_fn(which, model)
for B in [int(a) for a in sys.argv[2:]] or [1, 2]:
    noisy = torch.from_numpy(make_noisy(B, L, seed=7)).to(dev)
    for _ in range(3):
        eager = enhance(noisy)
    torch.cuda.synchronize()
    K = 20
    with collector_paused():
        t0 = time.perf_counter()
        for _ in range(K):
            eager = enhance(noisy)
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / K
    static_in = noisy.clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        enhance(static_in)  # warm-up on the capture stream
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.cuda.graph(g):
        static_out = enhance(static_in)
    static_in.copy_(noisy)
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(static_out, eager)
    with collector_paused():
        t0 = time.perf_counter()
        for _ in range(K):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / K
    print(f"{which} B={B}: eager {t_eager * 1e3:.2f} ms, graph replay {t_graph * 1e3:.2f} ms, bit-identical={same}")
