"""world_size-2 (and 3) gloo tests of the sharding / re-assembly logic used by bench.py --gpus N."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fullsubnet_amd.parallel import enhance_sharded, gather_shards, shard_bounds


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_items * 5, dtype=torch.float32).reshape(n_items, 5)
        fake_enhance = lambda x: x * 2 + 1  # stands in for Model.enhance (per-utterance independent)
        out = enhance_sharded(fake_enhance, full)
        assert torch.equal(out, full * 2 + 1), f"rank {rank}"
        lo, hi = shard_bounds(n_items, rank, world)
        g = gather_shards(full[lo:hi], n_items)
        assert torch.equal(g, full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 64), (2, 5), (3, 7)])
def test_sharded_enhance_reassembles_the_batch(world, n_items):
    mp.spawn(_worker, args=(world, _free_port(), n_items), nprocs=world, join=True)
