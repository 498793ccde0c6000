"""Multi-GPU layout of the path: one process per GPU, the batch x frequency rows are independent
sequences, so utterances are sharded across ranks with no data-path collective; the only exchange
is one all-gather that re-assembles the per-rank results (RCCL over xGMI with backend "nccl";
"gloo" in the CPU tests).  The C ABI itself stays collective-free."""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced split of ``n_items`` (first ``n_items % world`` ranks take one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_shards(local, n_items, group=None):
    """All-gather per-rank shards ``local [n_local, ...]`` (contiguous split of ``shard_bounds``)
    into the full ``[n_items, ...]`` tensor on every rank.  One collective; shards are padded to
    the largest shard so that a single all_gather_into_tensor moves everything."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    n_max = shard_bounds(n_items, 0, world)[1]
    send = local
    if local.shape[0] != n_max:
        send = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]].copy_(local)
    out = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)  # rank r's shard at rows [r n_max, ...)
    out = out.view((world, n_max) + tuple(local.shape[1:]))
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_items, r, world)
        parts.append(out[r, : hi - lo])
    return torch.cat(parts, dim=0)


def enhance_sharded(enhance_fn, noisy_full, group=None):
    """Run ``enhance_fn`` (e.g. ``Model.enhance``) on this rank's utterances of ``noisy_full [B, L]``
    and return the re-assembled ``[B, L]`` result on every rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(noisy_full.shape[0], rank, world)
    local = enhance_fn(noisy_full[lo:hi]) if hi > lo else noisy_full[:0]
    return gather_shards(local, noisy_full.shape[0], group=group)
