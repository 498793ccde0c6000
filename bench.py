#!/usr/bin/env python
"""bench.py - frames/s of the FullSubNet enhancement path (BASELINE.json configs[1]) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path (STFT -> FullSubNet -> cIRM decompress/apply -> iSTFT,
inferencer.py:130-145) over one batch of 64 synthetic 3 s / 16 kHz utterances already resident in
HBM.  With N > 1 ranks the 64 utterances are sharded across ranks (batch x frequency rows are
independent sequences), each rank runs the path on its shard and the enhanced waveforms are
re-assembled with one RCCL all-gather: total work is fixed -> "scaling": "strong".

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline     - fp32-MFMA roofline fraction of the dominant kernel (the sub-band recurrent kernel),
                 from its algorithmic FLOPs / its HIP-event duration on the launch stream
  cpu_baseline - the CPU oracle (numpy port of the reference path) timed on this box's host cores
                 on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, N_FFT, HOP = 16000, 512, 256
F, LA, NB, H_FB, H_SB = 257, 2, 15, 512, 384
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
# MAC per utterance per computed frame
MAC_FB = 2048 * (257 + 512) + 2048 * (512 + 512) + 512 * 257
MAC_SB_PER_BIN = 1536 * (32 + 384) + 1536 * (384 + 384) + 384 * 2
MAC_PER_FRAME = MAC_FB + 257 * MAC_SB_PER_BIN
MAC_REC_PER_ROW_STEP = 1536 * 384  # h_{t-1} W_hh^T of one sub-band LSTM layer


def build_model(device):
    import fullsubnet_amd
    from oracle.fullsubnet_oracle import make_params  # seeded weights only (not a compute path)
    params = make_params(seed=0, gain=2.0, mask_gain=24.0)
    model = fullsubnet_amd.Model(num_freqs=F, look_ahead=LA, sequence_model="LSTM", fb_num_neighbors=0,
                                 sb_num_neighbors=NB, fb_output_activate_function="ReLU",
                                 sb_output_activate_function=False, fb_model_hidden_size=H_FB,
                                 sb_model_hidden_size=H_SB, norm_type="offline_laplace_norm",
                                 num_groups_in_drop_band=1, weight_init=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return model.to(device).eval(), params


def cpu_baseline(params, length, budget_s=20.0):
    """Reference path restated on the CPU (oracle/, numpy + OpenBLAS on all host cores), timed on
    a bounded sample: whole utterances of the same shape, as many as fit ~budget_s."""
    from oracle import fullsubnet_oracle as O
    # the port scales to ~16 threads (MKL GEMMs of 257-row panels + numpy elementwise); more threads
    # only add contention on a 256-core host, so that is what is used and what `cores` reports
    cores = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(cores)
    win = torch.hann_window(N_FFT).numpy()
    frames_per_utt = 1 + length // HOP
    noisy = O.make_noisy(2, length, seed=77)
    O.full_band_crm_mask(noisy[:1], params, window=win)  # warm-up (thread pools, page faults)
    t0 = time.perf_counter()
    O.full_band_crm_mask(noisy, params, window=win)  # calibration
    per_utt = (time.perf_counter() - t0) / 2
    nb = int(max(2, min(64, budget_s // max(per_utt, 1e-3))))
    noisy = O.make_noisy(nb, length, seed=78)
    t0 = time.perf_counter()
    O.full_band_crm_mask(noisy, params, window=win)
    dt = time.perf_counter() - t0
    return {"value": round(nb * frames_per_utt / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{nb} x {length / SR:.1f} s utterance(s) in one batch, full path (oracle/fullsubnet_oracle.py: "
                      f"numpy {np.__version__} + torch-CPU/MKL GEMMs, {cores} threads), {dt:.1f} s wall",
            "rtf_speedup": round(nb * length / SR / dt, 3)}


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_hbm_traffic.json: FETCH_SIZE x2 (gfx950) + WRITE_SIZE at this exact config)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            return json.load(f)["dominant_kernel"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per step, whole job")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)  # RCCL on ROCm

    from fullsubnet_amd import _lib
    from fullsubnet_amd.parallel import shard_bounds
    from oracle.fullsubnet_oracle import make_noisy  # seeded synthetic input generator

    length = int(round(args.seconds * SR))
    T = 1 + length // HOP
    Tp = T + LA
    B = args.batch
    lo, hi = shard_bounds(B, rank, world)
    model, params = build_model(device)
    noisy_all = make_noisy(B, length, seed=1234)
    noisy = torch.from_numpy(noisy_all[lo:hi]).to(device)  # resident in HBM before the timed region
    b_loc = hi - lo
    b_max = shard_bounds(B, 0, world)[1]
    gathered = torch.empty((world * b_max, length), dtype=torch.float32, device=device) if world > 1 else None
    send = torch.zeros((b_max, length), dtype=torch.float32, device=device) if world > 1 else None

    def step():
        enh = model.enhance(noisy, n_fft=N_FFT, hop_length=HOP)
        if world > 1:
            send[:b_loc].copy_(enh)
            dist.all_gather_into_tensor(gathered, send)  # RCCL over xGMI: re-assemble the batch
            return gathered
        return enh

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    L = _lib.lib()
    L.fsn_profile_enable(1)  # hipEvents on the launch stream around every stage (no host sync inside)
    stage_ms = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in _lib.profile_read().items():  # waits for this step's events only
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    fence()
    dt = time.perf_counter() - t0
    L.fsn_profile_enable(0)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        frames = B * T
        value = frames * args.steps / dt
        stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
        # dominant kernel: lstm_rec_kernel, launched twice per step (sub-band layers 0 and 1)
        rec_ms = 0.5 * (stage_ms["sb_rec_l0"] + stage_ms["sb_rec_l1"])
        rec_flops = 2.0 * MAC_REC_PER_ROW_STEP * (b_loc * F) * Tp
        achieved = rec_flops / (rec_ms * 1e-3) / 1e12 if rec_ms > 0 else 0.0
        path_flops = 2.0 * MAC_PER_FRAME * b_loc * Tp
        out = {
            "metric": "frames/sec (16 kHz, 512-FFT, hop 256), whole job", "value": round(value, 1),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FullSubNet inference (full_band_crm_mask path), 16 kHz, n_fft 512, hop 256, "
                                   f"n_neighbour 15, look_ahead 2, batch {B} x {args.seconds:g} s, "
                                   f"offline_laplace_norm, full 257-bin mask per utterance",
                       "batch": B, "samples": length, "frames_per_utterance": T,
                       "parallelism": f"batch-shard x{world}" + (" + all-gather" if world > 1 else "")},
            "rtf_speedup_audio_s_per_s": round(value / (SR / HOP), 1),
            "rtf_classic": round((SR / HOP) / value, 6),
            "roofline": {"bound": "mfma", "kernel": "lstm_rec_kernel<384,RT,2> (sub-band recurrent, per layer)",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         # PMC passes are separate rocprofv3 runs at this exact config (B = 64, 1 GPU)
                         "traffic": measured_traffic() if (world == 1 and B == 64 and length == 48000) else None,
                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc)",
                         "flops_per_launch": rec_flops, "ms_per_launch": round(rec_ms, 3),
                         "whole_path_frac": round(path_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(params, length, args.cpu_budget)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None  # reported at N = 1 only
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
