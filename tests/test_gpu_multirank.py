"""Multi-rank readiness on a ONE-GPU box (recipes/dns_interspeech_2020/train.py:29, audio_zen/trainer/base_trainer.py:32:
one process per GPU).  The driver's 8-GPU scaling run is the only place RCCL runs across devices; what can be decided
on one GPU is decided here:

* `bench.py`'s own `world > 1` branch - sharding, padded all-gather, max-over-ranks timing, the other-mode side
  figure, the JSON line - launched exactly as the driver launches it (`python -m torch.distributed.run ...`) with two
  ranks sharing this GPU over gloo (RCCL refuses two ranks on one device) and the persistent kernels out of the plans
  (`--persistent never`: two PROCESSES on one GPU are the one situation the residency contract excludes);
* the persistent kernels INSIDE a two-rank job: the ranks take turns (device sync + barrier between the turns, so the
  chip belongs to one process at a time - what one process per GPU gives for free), each running its share on the
  persistent plans; the re-assembled result must equal the single-process call.

Needs an MI355X:  python -m pytest tests -m gpu"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import fullsubnet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODEL_KW = dict(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                sb_model_hidden_size=384, weight_init=False)


@pytest.fixture(scope="module")
def fsn():
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a ROCm device")
    import fullsubnet_amd
    fullsubnet_amd._lib.lib()
    return fullsubnet_amd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("shard,batch,extras", [("utterances", 6, True), ("utterances", 5, False), ("rows", 5, False),
                                                ("auto", 3, False)])
def test_bench_two_ranks_as_the_driver_launches_it(fsn, shard, batch, extras):
    """bench.py --gpus 2 under torch.distributed.run: strong scaling of ONE batch by whole utterances (even and uneven:
    the padded all-gather), by batch x frequency rows, and `auto` (rows when the utterances do not divide evenly)."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", str(batch), "--seconds", "1", "--shard", shard, "--persistent", "never",
           "--no-cpu-baseline"] + ([] if extras else ["--no-extras"])
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["unit"] == "frames/s" and out["dtype"] == "f32"
    T = 1 + 16000 // 256
    assert out["config"]["batch_total"] == batch and out["config"]["frames_per_utterance"] == T
    # whole-job value: the ONE batch's frames over the max-over-ranks time
    assert abs(out["value"] - batch * T / (out["ms_per_step"] * 1e-3)) <= 2e-3 * out["value"]
    rows = shard == "rows" or (shard == "auto" and batch % 2)
    assert ("row-shard x2" in out["config"]["parallelism"]) == bool(rows), out["config"]["parallelism"]
    assert out.get("cpu_baseline") is None  # reported at N = 1 only
    # the diagnosis block: every rank's own clock (never above the max-over-ranks figure), its all-gather time, its plan
    pr = out["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and all(0 < p["ms_per_step"] <= out["ms_per_step"] * 1.001 for p in pr)
    assert max(p["ms_per_step"] for p in pr) >= 0.999 * out["ms_per_step"]
    assert all(p["all_gather_ms"] >= 0 and p["plan"]["persistent_workgroups"] == 0 for p in pr)  # --persistent never
    assert (sum(p["utterances"] for p in pr) == batch) != bool(rows)  # row shard: every rank holds the whole batch
    assert out["roofline"]["rows"]["of"] > 0
    if extras:
        assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["batch_total"] == 2 * batch


def _turns_worker(rank, world, port, out_path):
    import torch.distributed as dist
    import fullsubnet_amd as fsn
    from fullsubnet_amd.parallel import gather_shards, shard_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        params = O.make_params(seed=0, gain=2.0, mask_gain=24.0)
        model = fsn.Model(norm_type="offline_laplace_norm", num_groups_in_drop_band=1, **MODEL_KW)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        model = model.cuda().eval()
        noisy = torch.from_numpy(O.make_noisy(16, 8192, seed=5)).cuda()
        lo, hi = shard_bounds(16, rank, world)   # 8 utterances per rank: group kernel + full-band chain (persistent)
        local = None
        for turn in range(world):                # one process on the chip at a time
            if turn == rank:
                before = fsn._lib.persist_stats()[0]
                local = model.enhance(noisy[lo:hi])
                torch.cuda.synchronize()
                assert fsn._lib.persist_stats()[0] - before >= 2, "the share must run on the persistent launches"
                assert fsn._lib.stream_status(synchronize=True) == (0, 0)
            torch.cuda.synchronize()
            dist.barrier()
        full = gather_shards(local, 16)
        if rank == 0:
            whole = torch.cat([model.enhance(noisy[:8]), model.enhance(noisy[8:])], dim=0)  # same plans, one process
            torch.cuda.synchronize()
            np.save(out_path, np.array([float((full - whole).abs().max()), float(whole.abs().max())]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_persistent_kernels_inside_a_two_rank_job(fsn, tmp_path):
    """Two ranks, 8 utterances each on the persistent plans (group kernel + full-band chain), taking turns on the
    shared GPU; the all-gathered batch is bit-equal to the same two calls issued by one process."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "turns.npy")
    mp.spawn(_turns_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    d, scale = np.load(out)
    assert scale > 0 and d == 0.0, (d, scale)
