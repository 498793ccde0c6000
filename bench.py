#!/usr/bin/env python
"""bench.py - frames/s of the FullSubNet enhancement path (BASELINE.json configs[1]) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path (STFT -> FullSubNet -> cIRM decompress/apply -> iSTFT,
inferencer.py:130-145) over a batch of synthetic 3 s / 16 kHz utterances already resident in HBM:
64 utterances per GPU (BASELINE config 2).  Utterances (and with them the batch x frequency rows of
the sub-band model) are independent, so with N ranks every rank runs the unchanged single-GPU path
on its own 64 utterances with no data-path collective, and one RCCL all-gather re-assembles the
enhanced node batch: per-GPU work is fixed -> "scaling": "weak" (the default).  `--scaling strong`
instead shards ONE 64-utterance batch across the ranks (8 utterances per rank at N = 8).

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline     - fp32-MFMA roofline fraction of the dominant kernel (the sub-band recurrent kernel,
                 two launches per step) from its algorithmic FLOPs / its HIP-event duration on the
                 launch stream; `traffic` from the committed rocprofv3 PMC passes of this config
  cpu_baseline - the CPU oracle (numpy/MKL port of the reference path) timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N = 1 only).
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
# the recurrent kernel, per sub-band row and step: layer 0 = W_hh h + the fused W_ih x (K = 32),
# layer 1 = W_hh h only (its K = 384 input projection is a separate GEMM)
MAC_REC_L0 = 1536 * (384 + 32)
MAC_REC_L1 = 1536 * 384


def build_model(device):
    import fullsubnet_amd
    from fsn_synthetic import make_params  # seeded weights
    params = make_params(seed=0, gain=2.0, mask_gain=24.0)
    model = fullsubnet_amd.Model(num_freqs=F, look_ahead=LA, sequence_model="LSTM", fb_num_neighbors=0,
                                 sb_num_neighbors=NB, fb_output_activate_function="ReLU",
                                 sb_output_activate_function=False, fb_model_hidden_size=H_FB,
                                 sb_model_hidden_size=H_SB, norm_type="offline_laplace_norm",
                                 num_groups_in_drop_band=1, weight_init=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return model.to(device).eval(), params


def cpu_baseline(params, length, budget_s=20.0):
    """Reference path restated on the CPU (oracle/), timed on a bounded sample: whole utterances of
    the same shape in one batch, as many as fit ~budget_s."""
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


def experimental_f16x3(args):
    """The same step with the opt-in split-precision kernels (FSN_F16X3=1: fp16 x 3 MFMAs with fp32 accumulation for
    both sub-band recurrent layers and the projection between them), measured in a child process because the switch
    is read once per process.  Reported NEXT TO the line's `value`, which is always the default fp32 build."""
    import subprocess
    if os.environ.get("FSN_F16X3") == "1":
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch), "--seconds", str(args.seconds), "--no-cpu-baseline", "--no-experimental"]
    try:
        out = subprocess.run(cmd, env=dict(os.environ, FSN_F16X3="1"), capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        return {"switch": "FSN_F16X3=1", "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                "stage_ms": d["stage_ms"],
                "note": "opt-in (fp32 operands split into two fp16 halves, three 16-bit MFMAs per product block, fp32 "
                        "accumulation); mask within 1.2e-5 of the fp32 path, GPU parity tests green with the switch on"}
    except Exception as e:  # the experiment must never break the benchmark line
        return {"switch": "FSN_F16X3=1", "error": str(e)[:200]}


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_hbm_traffic_end.json: FETCH_SIZE x2 (gfx950) + WRITE_SIZE at this exact config, mean of the
    two launches per step)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_end.json")) as f:
            return json.load(f)["dominant_kernel"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per step: per GPU (weak) / whole job (strong)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--shard", choices=["utterances", "rows"], default="utterances",
                    help="strong scaling only: whole utterances per rank (all-gather of waveforms), or contiguous slices "
                         "of the batch x frequency rows of the sub-band model (all-gather of the full-band mask; "
                         "balances any batch over any number of ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-experimental", action="store_true", help="skip the opt-in split-precision side measurement")
    ap.add_argument("--host-io", action="store_true",
                    help="side measurement for DESIGN.md: every step also copies its input from pinned host memory and "
                         "its result back (the PCIe-inclusive rate; never the headline `value`, which is HBM-resident)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback for the product path)"
    dev_index = local_rank % torch.cuda.device_count()  # one rank per GPU when launched as the contract says
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  BENCH_DIST_BACKEND=gloo exists only to exercise the multi-rank control
        # flow on a single-GPU box (ranks then share the device; RCCL refuses that).
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    from fullsubnet_amd import _lib
    from fullsubnet_amd.parallel import shard_bounds
    from fsn_synthetic import make_noisy  # seeded synthetic input

    length = int(round(args.seconds * SR))
    T = 1 + length // HOP
    Tp = T + LA
    if args.scaling == "weak":
        b_total, b_loc = args.batch * world, args.batch
        noisy_np = make_noisy(b_loc, length, seed=1234 + rank)
    elif args.shard == "rows":
        b_total = b_loc = args.batch  # every rank holds the batch; its share is a slice of the B F sub-band rows
        noisy_np = make_noisy(b_total, length, seed=1234)
    else:
        b_total = args.batch
        lo, hi = shard_bounds(b_total, rank, world)
        b_loc = hi - lo
        noisy_np = make_noisy(b_total, length, seed=1234)[lo:hi]
    b_max = shard_bounds(b_total, 0, world)[1]
    model, params = build_model(device)
    noisy = torch.from_numpy(noisy_np).to(device)  # resident in HBM before the timed region
    gathered = torch.empty((world * b_max, length), dtype=torch.float32, device=device) if world > 1 else None
    send = torch.zeros((b_max, length), dtype=torch.float32, device=device) if world > 1 else None

    if args.host_io:
        host_in = torch.from_numpy(noisy_np).pin_memory()
        host_out = torch.empty_like(host_in).pin_memory()

    row_sharded = args.scaling == "strong" and args.shard == "rows"
    if row_sharded:
        from fullsubnet_amd.parallel import enhance_row_sharded
        gathered = send = None

    def step():
        if args.host_io:
            noisy.copy_(host_in, non_blocking=True)
        if row_sharded:  # stft -> this rank's rows of the model -> all-gather of the mask -> decompress, apply, istft
            return enhance_row_sharded(model, noisy, N_FFT, HOP)
        enh = model.enhance(noisy, n_fft=N_FFT, hop_length=HOP)
        if args.host_io:
            host_out.copy_(enh, non_blocking=True)
        if world > 1:
            send[:b_loc].copy_(enh)
            dist.all_gather_into_tensor(gathered, send)  # RCCL over xGMI: re-assemble the node batch
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
        value = b_total * T * args.steps / dt
        stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
        # dominant kernel: lstm_rec_kernel, launched twice per step (sub-band layers 0 and 1)
        rows_loc = shard_bounds(b_total * F, 0, world)[1] if row_sharded else b_loc * F
        rows_steps = float(rows_loc) * Tp
        rec_flops = 2.0 * (MAC_REC_L0 + MAC_REC_L1) * rows_steps  # both launches
        rec_ms = stage_ms["sb_rec_l0"] + stage_ms["sb_rec_l1"]
        achieved = rec_flops / (rec_ms * 1e-3) / 1e12 if rec_ms > 0 else 0.0
        path_flops = 2.0 * (MAC_FB * b_loc + MAC_SB_PER_BIN * rows_loc) * Tp
        at_config2 = world == 1 and b_loc == 64 and length == 48000
        out = {
            "metric": "frames/sec (16 kHz, 512-FFT, hop 256), whole job", "value": round(value, 1),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (host buffers in and out over PCIe every step: --host-io)" if args.host_io else ""),
            "config": {"workload": f"FullSubNet inference (full_band_crm_mask path), 16 kHz, n_fft 512, hop 256, "
                                   f"n_neighbour 15, look_ahead 2, batch {b_loc} x {args.seconds:g} s per GPU "
                                   f"({b_total} utterances per step in total), offline_laplace_norm, full 257-bin "
                                   f"mask per utterance",
                       "batch_per_gpu": b_loc, "batch_total": b_total, "samples": length,
                       "frames_per_utterance": T,
                       "parallelism": (f"row-shard x{world} ({rows_loc} of {b_total * F} sub-band rows per rank) + "
                                       f"all-gather of the mask" if row_sharded else
                                       f"utterance-shard x{world}" + (" + all-gather" if world > 1 else ""))},
            "rtf_speedup_audio_s_per_s": round(value / (SR / HOP), 1),
            "rtf_classic": round((SR / HOP) / value, 6),
            "roofline": {"bound": "mfma", "kernel": "lstm_rec_kernel<384,RT,2,*> (sub-band recurrent; 2 launches/step)",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         # PMC passes are separate rocprofv3 runs at config 2 (B = 64, 1 GPU)
                         "traffic": measured_traffic() if at_config2 else None,
                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc)",
                         "flops_per_launch": rec_flops / 2, "ms_per_launch": round(rec_ms / 2, 3),
                         "whole_path_frac": round(path_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        }
        if not args.no_experimental and world == 1 and os.environ.get("FSN_F16X3") != "1":
            torch.cuda.synchronize()
            out["experimental_f16x3"] = experimental_f16x3(args)
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
