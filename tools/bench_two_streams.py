"""Throughput mode: K batches alternating over two caller streams (two calls in flight) against the same K batches on one
stream - does batch i + 1's latency-bound front (full-band chain, block-pair wavefronts) run under batch i's persistent
sub-band kernels?  The library is re-entrant across streams (include/fsn_hip.h); outputs must be bit-identical to the serial
calls.  usage: bench_two_streams.py [fullsubnet|fast|improved48] [batch] [K]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_family as BF  # noqa: E402
from fsn_synthetic import make_noisy  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fast"
B = int(sys.argv[2]) if len(sys.argv) > 2 else {"fast": 256, "improved48": 32, "fullsubnet": 64}[which]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if which == "fullsubnet":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as BM
    model, _ = BM.build_model(torch.device("cuda"))
    L = 48000
    enhance = lambda y: model.enhance(y)
else:
    pack = BF.build(which)
    model, L = pack[0], pack[2]
    enhance = BF.enhance_fn(which, model)
batches = [torch.from_numpy(make_noisy(min(B, 8), L, seed=10 + i)).cuda().repeat((B + 7) // 8, 1)[:B].contiguous() for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
with torch.no_grad():
    ref = [enhance(b) for b in batches]
    torch.cuda.synchronize()

    def serial():
        outs = [enhance(batches[i % 2]) for i in range(K)]
        torch.cuda.synchronize()
        return outs

    def two():
        outs = []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for i in range(K):
            with torch.cuda.stream(streams[i % 2]):
                outs.append(enhance(batches[i % 2]))
        torch.cuda.synchronize()
        return outs

    res = {}
    for name, fn in (("one stream", serial), ("two streams", two), ("one stream", serial), ("two streams", two)):
        fn()
        t0 = time.perf_counter()
        outs = fn()
        dt = (time.perf_counter() - t0) / K * 1e3
        same = all(torch.equal(o, ref[i % 2]) for i, o in enumerate(outs))
        res.setdefault(name, []).append(dt)
        print(f"{which} B={B} K={K} {name}: {dt:.2f} ms per batch, bit-identical to the serial calls: {same}", flush=True)
