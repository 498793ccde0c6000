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


def pack_ragged(parts, n_flat):
    """Flatten a list of tensors into one zero-padded 1-D buffer of ``n_flat`` elements."""
    ref = parts[0]
    buf = torch.zeros(n_flat, dtype=ref.dtype, device=ref.device)
    o = 0
    for p in parts:
        buf[o:o + p.numel()].copy_(p.reshape(-1))
        o += p.numel()
    return buf


def gather_ragged(parts, n_items, group=None):
    """Several differently shaped tensors, each sharded along its leading axis, re-assembled with ONE collective.

    ``parts[i]`` is this rank's shard ``[n_local_i, *trailing_i]`` of a tensor with ``n_items[i]`` leading entries
    (contiguous split of ``shard_bounds(n_items[i], rank, world)``; a rank may own nothing of a short axis).  Every
    rank packs its shards into one flat buffer padded to the largest rank's size, a single all_gather_into_tensor
    moves everything (few large collectives suit the point-to-point xGMI links better than one per tensor), and the
    full tensors ``[n_items[i], *trailing_i]`` are cut out of the result on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return list(parts)
    trailing = [tuple(p.shape[1:]) for p in parts]
    row = [int(torch.Size(t).numel()) for t in trailing]

    def flat_size(r):
        return sum((shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0]) * w for n, w in zip(n_items, row))

    n_flat = max(flat_size(r) for r in range(world))
    send = pack_ragged(parts, n_flat)
    out = torch.empty(world * n_flat, dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)
    out = out.view(world, n_flat)
    full = []
    offsets = [0] * world
    for n, w, t in zip(n_items, row, trailing):
        pieces = []
        for r in range(world):
            lo, hi = shard_bounds(n, r, world)
            pieces.append(out[r, offsets[r]:offsets[r] + (hi - lo) * w].view((hi - lo,) + t))
            offsets[r] += (hi - lo) * w
        full.append(torch.cat(pieces, dim=0))
    return full


def enhance_row_sharded(model, noisy, n_fft=512, hop_length=256, group=None):
    """inferencer.py:130-145 with the batch x frequency rows of the sub-band model sharded over the group (SURVEY
    8e): every rank holds the whole ``noisy [B, L]``; the transforms and the mask application are cheap and run
    replicated, the model runs on this rank's contiguous share of the B F rows (``Model.forward_rows``: the full-band
    model only for the utterances that share touches), and ONE all-gather re-assembles the full-band mask
    [B, 2, F, T] before it is decompressed and applied.  Use it when there are fewer utterances than ranks or the
    batch does not divide evenly; with whole utterances per rank ``enhance_sharded`` does the same work with a
    smaller all-gather (waveforms instead of masks)."""
    from .acoustics.feature import istft, stft
    from .acoustics.mask import decompress_cIRM
    mag, _, re, im = stft(noisy, n_fft, hop_length, n_fft, return_phase=False)
    crm = model.forward_row_sharded(mag.unsqueeze(1), group=group)  # [B, 2, F, T]
    m = decompress_cIRM(crm.permute(0, 2, 3, 1))
    return istft((m[..., 0] * re - m[..., 1] * im, m[..., 1] * re + m[..., 0] * im), n_fft, hop_length, n_fft,
                 length=noisy.size(-1), input_type="real_imag")
