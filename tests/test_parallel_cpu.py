"""world_size-2 / 3 / 8 gloo tests of the sharding / re-assembly logic used by bench.py --gpus N (8 = the node the north star
names: config 2's 64 utterances and 16 448 batch x frequency rows, config 5's 55 unequal units)."""
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


@pytest.mark.parametrize("world,n_items", [(2, 64), (2, 5), (3, 7), (8, 64), (8, 5)])
def test_sharded_enhance_reassembles_the_batch(world, n_items):
    mp.spawn(_worker, args=(world, _free_port(), n_items), nprocs=world, join=True)


# ---- frequency-axis (sub-band unit) shard of Improved FullSubNet (BASELINE config 5) -------------------------------

def _fake_section_model(c):
    """Stands in for SubBandSequenceWrapper (which only runs on the GPU library): per-unit independent, like the
    real one - [B, N, 1, F_sub, T] -> [B, 2, N c, T]."""
    def run(x):
        B, N, _, Fs, T = x.shape
        a = x[:, :, 0, :c, :] * 2.0 + x[:, :, 0, Fs - c:, :]          # [B, N, c, T]
        b = x[:, :, 0, :c, :].cumsum(-1) - x[:, :, 0, 1:c + 1, :]
        return torch.stack([a, b], dim=1).reshape(B, 2, N * c, T)     # [B, 2, N, c, T] -> [B, 2, N c, T]
    return run


def _subband_model(num_freqs, cutoffs, centres, norm):
    from fullsubnet_amd.improved_fullsubnet import SubbandModel
    m = SubbandModel(freq_cutoffs=cutoffs, sb_num_center_freqs=centres, sb_num_neighbor_freqs=[15] * len(centres),
                     fb_num_center_freqs=centres, fb_num_neighbor_freqs=[15] * len(centres), sequence_model="LSTM",
                     hidden_size=64, norm_type=norm)
    del m.sb_models  # replaced by CPU stand-ins (the real ones only run on the HIP library)
    m.sb_models = [_fake_section_model(c) for c in centres]
    return m


_UNIT_CASES = [
    (480, [20, 120, 240], [1, 4, 20, 60], "offline_laplace_norm"),      # BASELINE config 5 (48 kHz): 20 + 25 + 6 + 4 = 55 units
    (256, [20, 80], [1, 4, 8], "offline_laplace_norm"),                 # 16 kHz default: 20 + 15 + 22 units
    (480, [32, 128, 192], [1, 4, 8, 16], "offline_laplace_norm"),       # a four-section layout, uneven over 3 ranks
    (64, [32], [1, 16], "offline_gaussian_norm"),                       # 2 units in the last section: ranks with none
]


def _unit_worker(rank, world, port, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        num_freqs, cutoffs, centres, norm = case
        g = torch.Generator().manual_seed(5)
        noisy = torch.rand(2, 1, num_freqs, 7, generator=g) + 0.1
        fb = torch.rand(2, 1, num_freqs, 7, generator=g) + 0.1
        m = _subband_model(num_freqs, cutoffs, centres, norm)
        with torch.no_grad():
            whole = m(noisy, fb)
            sharded = m(noisy, fb, unit_group=True)
        assert whole.shape == (2, 2, num_freqs, 7)
        assert torch.equal(whole, sharded), f"rank {rank}"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", _UNIT_CASES[1:], ids=["16k", "four_sections", "fewer_units_than_ranks"])
def test_unit_sharded_subband_model_equals_the_unsharded_one(world, case):
    mp.spawn(_unit_worker, args=(world, _free_port(), case), nprocs=world, join=True)


def test_unit_shard_of_config5_over_eight_ranks():
    """BASELINE config 5 on the node the north star names: 55 units of four unequal sections over 8 ranks (sections of 6 and 4
    units leave ranks without a unit of theirs), one ragged all-gather."""
    assert [shard_bounds(n, 7, 8) for n in (20, 25, 6, 4)] == [(18, 20), (22, 25), (6, 6), (4, 4)]
    mp.spawn(_unit_worker, args=(8, _free_port(), _UNIT_CASES[0]), nprocs=8, join=True)


def test_unit_shard_refuses_to_build_an_autograd_graph():
    m = _subband_model(64, [32], [1, 16], "offline_laplace_norm")
    x = torch.rand(1, 1, 64, 3) + 0.1
    with pytest.raises(RuntimeError, match="inference layout"):
        m(x, x, unit_group=True)


def test_gather_ragged_single_process_layout():
    """The packing used by the unit shard, without a process group: emulate every rank's buffer by hand."""
    from fullsubnet_amd.parallel import pack_ragged
    n_items, world = [5, 2, 7], 3
    full = [torch.arange(n * 6, dtype=torch.float32).reshape(n, 2, 3) + 100 * i for i, n in enumerate(n_items)]
    bufs = []
    for r in range(world):
        parts = [f[slice(*shard_bounds(n, r, world))] for f, n in zip(full, n_items)]
        bufs.append(pack_ragged(parts, 64))
        assert bufs[-1].numel() == 64
        assert bufs[-1][sum(p.numel() for p in parts):].abs().sum() == 0
    # rank 2 owns nothing of the 2-entry tensor, so its second tensor's data follows the first directly
    lo, hi = shard_bounds(5, 2, world)
    assert torch.equal(bufs[2][: (hi - lo) * 6].view(hi - lo, 2, 3), full[0][lo:hi])
    assert shard_bounds(2, 2, world) == (2, 2)


# ---- batch x frequency row shard of the fused FullSubNet forward ---------------------------------------

def _row_worker(rank, world, port, B, F, T):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fullsubnet_amd
        kw = dict(num_freqs=F, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                  fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=64,
                  sb_model_hidden_size=384, weight_init=False)
        m = fullsubnet_amd.Model(**kw)
        mag = torch.rand(B, 1, F, T, generator=torch.Generator().manual_seed(3))
        calls = []

        def fake_rows(noisy_mag, lo, hi):  # stands in for the HIP entry: row n = b F + f -> [2, T], per-row independent
            calls.append((lo, hi))
            rows = noisy_mag[:, 0].reshape(B * F, T)[lo:hi]
            return torch.stack([rows * 2 + 1, rows.flip(-1)], dim=1)

        m.forward_rows = fake_rows
        got = m.forward_row_sharded(mag)
        want = torch.stack([mag[:, 0] * 2 + 1, mag[:, 0].flip(-1)], dim=1)  # [B, 2, F, T]
        assert torch.equal(got, want), f"rank {rank}"
        assert calls == ([shard_bounds(B * F, rank, world)] if shard_bounds(B * F, rank, world)[1] > shard_bounds(B * F, rank, world)[0] else [])
    finally:
        dist.destroy_process_group()


# last of the small ones: a rank without rows; (8, 64, 257): config 2 on 8 ranks, 16 448 = 8 x 2056 rows
@pytest.mark.parametrize("world,B,F", [(2, 3, 257), (3, 1, 257), (3, 2, 5), (3, 1, 2), (8, 64, 257), (8, 9, 257)])
def test_row_sharded_forward_reassembles_the_mask(world, B, F):
    mp.spawn(_row_worker, args=(world, _free_port(), B, F, 6), nprocs=world, join=True)
