#!/usr/bin/env python
"""bench.py - frames/s of the FullSubNet enhancement path (BASELINE.json configs[1]) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path (STFT -> FullSubNet -> cIRM decompress/apply -> iSTFT,
inferencer.py:130-145) over a batch of synthetic 3 s / 16 kHz utterances already resident in HBM:
one batch of 64 utterances (BASELINE config 2).  Utterances (and with them the batch x frequency rows of the
sub-band model) are independent, so with N ranks that ONE batch is sharded - 64 / N whole utterances per rank (or
contiguous slices of the B F sub-band rows, --shard rows) - every rank runs the unchanged single-GPU path on its
share with no data-path collective, and one RCCL all-gather re-assembles the node batch: total work is fixed ->
"scaling": "strong", the north-star target (>= 6x at 8 GPUs).  The weak form (64 utterances per rank,
`--scaling weak`) is measured next to it as the side figure `other_scaling`.

Prints ONE JSON line (rank 0) with the driver's contract keys plus
  roofline     - fp32-MFMA roofline fraction of the dominant kernel (the sub-band recurrent kernel,
                 two launches per step) from its algorithmic FLOPs / its HIP-event duration on the
                 launch stream; `traffic` / `mfma_busy_frac` from the committed rocprofv3 PMC passes of this config
  cpu_baseline - the reference's ATen operator sequence (oracle/aten_baseline.py) timed on this box's host cores
                 on a bounded sample of the same workload (rank 0, N = 1 only); the numpy oracle as `oracle_port`
  one_utterance (eager / hipGraph replay latency of one utterance), split_f16x3, train_step, fast_b256, fast_train_b72_amp, improved48_b32 -
  side figures (N = 1): the opt-in split-precision
                 kernels, one training step at BASELINE config 3's per-rank shape, Fast FullSubNet at batch 256
                 (config 4) and Improved FullSubNet at 48 kHz, batch 32 (config 5).
"""
import argparse
import contextlib
import gc
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
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec (6290 GB/s measured copy)
# MAC per utterance per computed frame
MAC_FB = 2048 * (257 + 512) + 2048 * (512 + 512) + 512 * 257
MAC_SB_PER_BIN = 1536 * (32 + 384) + 1536 * (384 + 384) + 384 * 2
MAC_PER_FRAME = MAC_FB + 257 * MAC_SB_PER_BIN
# the two persistent recurrent kernels, per sub-band row and step: layer 0 (lstm_rec_in_kernel) = W_hh h + W_ih x with
# K = 32 + 384; layer 1 (lstm_rec_x_kernel) = W_ih h0 + W_hh h with K = 384 + 384 (its input projection is inside
# since round 2: no projection GEMM, no gx round trip)
MAC_REC_L0 = 1536 * (384 + 32)
MAC_REC_L1 = 1536 * (384 + 384)


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


def _time_aten(model, length, batch, threads):
    from fsn_synthetic import make_noisy
    from oracle import aten_baseline as A
    torch.set_num_threads(threads)
    noisy = torch.from_numpy(make_noisy(batch, length, seed=78))
    t0 = time.perf_counter()
    A.full_band_crm_mask(model, noisy)
    return time.perf_counter() - t0


def _all_cores_sample(avail, cap_s=45.0, seconds=0.5):
    """SURVEY 8(d)'s figure with EVERY host thread the process may use (n = len(os.sched_getaffinity(0)), stated), on a
    reduced sample so that it fits the default run: ONE utterance of `seconds` of audio through the same ATen operator
    sequence, in a child process with a hard time limit (oneDNN's LSTM collapses when oversubscribed - r03: 5 frames/s
    on 4 utterances with 256 threads - and a running ATen call cannot be interrupted from inside).  A run that hits the
    limit is reported as an upper bound."""
    import subprocess
    n = int(round(seconds * SR))
    code = (
        "import sys, time, torch; sys.path.insert(0, %r)\n"
        "from fsn_synthetic import make_noisy, make_params\n"
        "from oracle import aten_baseline as A\n"
        "torch.set_num_threads(%d)\n"
        "m = A.AtenFullSubNet(make_params(seed=0, gain=2.0, mask_gain=24.0)).eval()\n"
        "x = torch.from_numpy(make_noisy(1, %d, seed=78))\n"
        "A.full_band_crm_mask(m, x[:, :2048])\n"
        "t0 = time.perf_counter(); A.full_band_crm_mask(m, x); print(time.perf_counter() - t0)\n" % (ROOT, avail, n))
    frames = 1 + n // HOP
    sample = f"1 x {seconds:g} s ({frames} frames), the ATen operator sequence with all {avail} host threads"
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=cap_s)
        dt = float(r.stdout.strip().splitlines()[-1])
        return {"value": round(frames / dt, 2), "unit": "frames/s", "cores": avail, "sample": sample + f", {dt:.1f} s wall"}
    except subprocess.TimeoutExpired:
        return {"value": round(frames / cap_s, 2), "unit": "frames/s", "cores": avail, "upper_bound": True,
                "sample": sample + f": not finished after {cap_s:g} s (the figure is an upper bound)"}
    except Exception as e:  # a side figure must never break the line
        return {"value": None, "cores": avail, "error": str(e)[:160], "s": round(time.perf_counter() - t0, 1)}


def cpu_baseline(params, length, budget_s=30.0, all_cores_too=False):
    """The reference's CPU arithmetic on this box's host cores, on a bounded sample of the same workload.

    Headline figure (`value`): oracle/aten_baseline.py - the ATen operator sequence the reference itself executes
    (torch.stft, nn.LSTM(257,512,2) + Linear + ReLU, F.unfold, nn.LSTM(32,384,2) + Linear, decompress, torch.istft;
    oneDNN / MKL kernels of the PyTorch build on this box) - whole 3 s utterances in ONE batch (up to config 2's 64,
    as many as the budget allows), timed at several thread counts ON THAT BATCH: `value` is the fastest, `all_cores`
    the figure with every host thread the process may use (n = len(os.sched_getaffinity(0))).
    Second, labelled figure (`oracle_port`): the numpy restatement that the parity tests use as their checker."""
    from oracle import aten_baseline as A
    from oracle import fullsubnet_oracle as O
    avail = len(os.sched_getaffinity(0))
    frames_per_utt = 1 + length // HOP
    model = A.AtenFullSubNet(params).eval()
    _time_aten(model, length, 1, min(avail, 16))  # warm-up: thread pools, oneDNN primitive cache, page faults
    t1 = _time_aten(model, length, 1, min(avail, 16))
    # a batch costs ~0.35 of its utterances run one by one (measured); three timed passes share the budget
    nb = int(max(1, min(64, (budget_s / 3.0) // max(0.35 * t1, 1e-3))))
    cal, spent = {}, 0.0
    for th in sorted({min(avail, t) for t in (16, 32, 64)}):
        if cal and spent > 0.8 * budget_s:  # never skip the first figure; later ones only while the budget lasts
            break
        cal[th] = _time_aten(model, length, nb, th)
        spent += cal[th]
    threads = min(cal, key=cal.get)
    dt = cal[threads]
    # every host thread the process may use (n = len(os.sched_getaffinity(0))): on the 256-thread GPU box
    # oneDNN's LSTM collapses when oversubscribed - measured in r03: 66 frames/s on the 64-utterance batch (182 s of wall)
    # and 5 frames/s on 4 utterances (150 s), against 1310 - 1356 at 16 threads - so the figure is opt-in
    # (--cpu-all-cores); the default run reports the scaling over 16 / 32 / 64 threads instead
    all_cores = None
    if avail in cal:
        all_cores = {"value": round(nb * frames_per_utt / cal[avail], 2), "unit": "frames/s", "cores": avail}
    elif all_cores_too:
        t_all = _time_aten(model, length, nb, avail)
        all_cores = {"value": round(nb * frames_per_utt / t_all, 2), "unit": "frames/s", "cores": avail,
                     "sample": f"the same {nb}-utterance batch, {t_all:.1f} s wall"}
    else:
        all_cores = _all_cores_sample(avail)
    out = {"value": round(nb * frames_per_utt / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{nb} x {length / SR:.1f} s utterance(s) in one batch, full path stft -> model -> decompress -> "
                     f"mask -> istft as the ATen operator sequence of the reference (oracle/aten_baseline.py: torch "
                     f"{torch.__version__} CPU kernels, oneDNN LSTM + MKL FFT), {threads} of {avail} host threads "
                     f"(fastest of {sorted(cal)}, each timed on this same batch), {dt:.1f} s wall",
           "rtf_speedup": round(nb * length / SR / dt, 3),
           "by_threads": {str(th): round(nb * frames_per_utt / t, 2) for th, t in sorted(cal.items())},
           "all_cores": all_cores,
           "all_cores_note": f"n = len(os.sched_getaffinity(0)) = {avail}; oneDNN's LSTM slows down when oversubscribed (the whole "
                             f"64-utterance batch with all 256 threads: 66 frames/s in r03, profiles/r03_cpu_threads.md; "
                             f"--cpu-all-cores times that batch), which is why `value` is the fastest of 16 / 32 / 64 threads"}
    # the parity checker (numpy + torch-CPU matmuls), for the record: it scales to ~16 threads
    cores = min(16, avail)
    torch.set_num_threads(cores)
    win = torch.hann_window(N_FFT).numpy()
    noisy = O.make_noisy(2, length, seed=77)
    O.full_band_crm_mask(noisy[:1], params, window=win)
    t0 = time.perf_counter()
    ref, inter = O.full_band_crm_mask(noisy, params, window=win, return_intermediates=True)
    dt = time.perf_counter() - t0
    out["_oracle_outputs"] = (noisy, ref, inter["crm"])  # popped by main(): the checker's side of `parity`
    out["oracle_port"] = {"value": round(2 * frames_per_utt / dt, 2), "unit": "frames/s", "cores": cores,
                          "sample": f"2 x {length / SR:.1f} s, oracle/fullsubnet_oracle.py (numpy {np.__version__} + "
                                    f"torch-CPU matmuls), {dt:.1f} s wall"}
    return out


@contextlib.contextmanager
def collector_paused():
    """CPython's cyclic collector collected before and held during a timed region: a generation-2 pass over torch's
    ~10^6 objects is a ~40 ms host pause that otherwise lands in one region or another at random (found with the HIP API
    trace on the 4.7 ms step of config 5 at one utterance)."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def timed_steps(step, fence, steps, read_profile=None):
    """K steps bracketed by fence() on both sides; returns (seconds, summed stage ms)."""
    stage_ms = {}
    fence()
    with collector_paused():  # a generation-2 pass of the cyclic collector is a ~40 ms host pause (tools/bench_family.py)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            if read_profile is not None:
                for k, v in read_profile().items():  # waits for this step's events only
                    stage_ms[k] = stage_ms.get(k, 0.0) + v
        fence()
        dt = time.perf_counter() - t0
    return dt, stage_ms


def amp_step_traffic(saves):
    """HBM bytes of ONE autocast training step REPLAYED from the committed PMC passes (tools/gpu_run_pmc_train.sh:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of tools/bench_train.py 16 f16, summed over every kernel of a
    step, FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) with the stamp of the sources they were taken on."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rocprof_pmc import source_stamp
    name = "r06_pmc_train_f16.json" if saves == "16" else "r06_pmc_train_f16_saves32.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            j = json.load(f)
        total = sum(k.get("hbm_bytes_per_launch", 0.0) * k["dispatches"] for k in j["kernels"].values()) / 5.0  # 2 warm-up + 3 timed steps
        top = sorted(j["kernels"].items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0.0) * kv[1]["dispatches"])[:2]
        built = j.get("build") or {}
        return {"bytes_per_step": total, "src": "profiles/" + name, "taken_on": built,
                "stale": built.get("csrc_sha256") != source_stamp(ROOT)["csrc_sha256"] if built else None,
                "largest": {n.split("<")[0]: round(k["hbm_bytes_per_launch"] / 1e9, 2) for n, k in top}}
    except (OSError, KeyError, ValueError):
        return None


def amp_step_algorithmic_bytes(saves, rows=2048, steps=195, H=384):
    """What one autocast step of the sub-band model must move through HBM if every tensor is written once and read once per
    consuming launch (N rows, T' steps, G = 4H, s = bytes of a saved gate): forward writes both hidden sequences,
    the gate saves and the cell sequences N T' (8H + 2sG + 8H); BPTT reads the saves, the cell sequences and dH1 and writes
    the 16-bit gate gradients N T' (2sG + 8H + 4H + 4G); the products read those five times, convert and read the hidden
    sequences N T' (10G + 18H).  The exchanges between the workgroups of a persistent launch are NOT in it."""
    G, s = 4 * H, (2 if saves == "16" else 4)
    return float(rows) * steps * ((4 * s + 14) * G + 46 * H)


def training_step_ms(device, steps=5, arith="f32", saves=None):
    """Side figure (BASELINE config 3, per-rank shape): one step of fullsubnet/trainer.py:41-71 - 16 utterances x
    49 152 samples, drop_band groups 2, MSE on the compressed cIRM, clip_grad_norm_(10) + Adam, one GPU.
    arith "f32": use_amp = false.  "f16": the reference's own mode (train.toml:5 use_amp = true): autocast arithmetic
    (16-bit matrix-core operands, fp32 accumulation) with torch.amp.GradScaler around the fused optimizer."""
    import fullsubnet_amd
    from fullsubnet_amd.train import train_step
    from fsn_synthetic import make_noisy, make_params
    model = fullsubnet_amd.Model(num_freqs=F, look_ahead=LA, sequence_model="LSTM", fb_num_neighbors=0,
                                 sb_num_neighbors=NB, fb_output_activate_function="ReLU",
                                 sb_output_activate_function=False, fb_model_hidden_size=H_FB,
                                 sb_model_hidden_size=H_SB, norm_type="offline_laplace_norm",
                                 num_groups_in_drop_band=2, weight_init=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()})
    model = model.to(device).train()
    model.train_arithmetic = arith
    if saves is not None:
        model.train_saves = saves
    scaler = torch.amp.GradScaler("cuda", enabled=arith != "f32")
    opt = fullsubnet_amd.ClipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
    noisy = torch.from_numpy(make_noisy(16, 49152, seed=41)).to(device)
    clean = torch.from_numpy(0.7 * make_noisy(16, 49152, seed=42)).to(device)
    check = training_parity(device, arith, saves)
    for _ in range(2):
        train_step(model, opt, noisy, clean, scaler=scaler)
    torch.cuda.synchronize()
    with collector_paused():
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = train_step(model, opt, noisy, clean, scaler=scaler)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
    T = 1 + 49152 // HOP
    flops = 3 * 2.0 * 16 * (T + LA) * (MAC_FB + 128 * MAC_SB_PER_BIN)  # ~3x forward, 128 bins kept
    skipped = opt.skipped_steps()
    del model, opt
    torch.cuda.empty_cache()
    out = {"ms_per_step": round(ms, 2), "dtype": arith,
           "config": "BASELINE config 3 per-rank shape: 16 x 49152 samples, drop_band groups 2, cIRM MSE + "
                     "clip_grad_norm_(10) + Adam" + ("" if arith == "f32" else
                                                     f", use_amp = true: {arith} matrix-core operands with fp32 accumulation "
                                                     f"on the sub-band kernels + torch.amp.GradScaler (scale "
                                                     f"{scaler.get_scale():g}, {skipped} skipped updates)"),
           "loss": round(float(loss), 6), "tflops": round(flops / (ms * 1e-3) / 1e12, 1), **check}
    if arith == "f32":
        out["frac_fp32_mfma_peak"] = round(flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3)
    else:
        from fullsubnet_amd.train import DEFAULT_TRAIN_SAVES
        mode = saves or DEFAULT_TRAIN_SAVES
        out["saved_gates"] = f"{mode}-bit" + (" (Model.train_saves default)" if saves is None else f' (Model.train_saves = "{saves}")')
        tr = amp_step_traffic(mode)
        alg = amp_step_algorithmic_bytes(mode)
        if tr:
            gbs = tr["bytes_per_step"] / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(gbs / PEAK_HBM_GBS, 4), "frac_of_measured_copy_bandwidth": round(gbs / 6290.0, 4),
                               "traffic": tr["bytes_per_step"], "traffic_unit": "HBM bytes per step, all kernels (rocprofv3 --pmc)",
                               "traffic_source": f"replayed from {tr['src']}, NOT measured in this run", "pmc_taken_on": tr["taken_on"],
                               "pmc_stale": tr["stale"], "largest_launches_gb": tr["largest"],
                               "algorithmic_bytes": alg, "traffic_over_algorithmic": round(tr["bytes_per_step"] / alg, 2),
                               "note": "the step is bound by the memory system, not by the 16-bit matrix peak (2.5 PFLOP/s: "
                                       f"{round(flops / (ms * 1e-3) / 1e12 / 2500.0, 3)} of it)"}
    return out


def training_step_shipped_batches(device, steps=3):
    """The batch sizes the shipped training TOMLs say (fullsubnet/train.toml:52 batch_size = 32, train_cumulativeLaplaceNorm.toml:52
    batch_size = 48, both use_amp = true) on one GPU: the sub-band rows as pieces of 2048 through the persistent launches."""
    import fullsubnet_amd
    from fullsubnet_amd.train import train_step
    from fsn_synthetic import make_noisy, make_params
    out = {}
    for key, B, norm in (("b32_offline", 32, "offline_laplace_norm"), ("b48_cumulative", 48, "cumulative_laplace_norm")):
        model = fullsubnet_amd.Model(num_freqs=F, look_ahead=LA, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=NB,
                                     fb_output_activate_function="ReLU", sb_output_activate_function=False,
                                     fb_model_hidden_size=H_FB, sb_model_hidden_size=H_SB, norm_type=norm,
                                     num_groups_in_drop_band=2, weight_init=False)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()})
        model = model.to(device).train()
        model.train_arithmetic = "f16"
        scaler = torch.amp.GradScaler("cuda")
        opt = fullsubnet_amd.ClipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
        noisy = torch.from_numpy(make_noisy(B, 49152, seed=41)).to(device)
        clean = torch.from_numpy(0.7 * make_noisy(B, 49152, seed=42)).to(device)
        for _ in range(2):
            train_step(model, opt, noisy, clean, scaler=scaler)
        torch.cuda.synchronize()
        with collector_paused():
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = train_step(model, opt, noisy, clean, scaler=scaler)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
        T = 1 + 49152 // HOP
        flops = 3 * 2.0 * B * (T + LA) * (MAC_FB + 128 * MAC_SB_PER_BIN)
        out[key] = {"batch": B, "norm_type": norm, "ms_per_step": round(ms, 2), "loss": round(float(loss), 6),
                    "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "skipped_steps": opt.skipped_steps(),
                    "utterances_per_s": round(B / (ms * 1e-3), 1)}
        del model, opt, noisy, clean
        torch.cuda.empty_cache()
    out["note"] = ("use_amp = true (f16 operands on the sub-band kernels, GradScaler), 49152 samples per utterance; parity at 32 utterances: "
                   "tests/test_gpu_train.py::test_train_step_vs_reference_and_oracle[fsn_train_c3x2], tests/test_gpu_amp.py::"
                   "test_amp_step_at_the_shipped_batch_of_32")
    return out


def path_parity(model, oracle_outputs, batch, length, device):
    """The margin the timed arithmetic leaves under the north-star bound, in the line itself: the two utterances the
    numpy oracle just computed (cpu_baseline's `oracle_port` leg) sit at both ends of a batch of the TIMED size, so the
    plan - the persistent kernels with the hardware-transcendental gate functions (v_exp_f32 / v_rcp_f32,
    fsn_common.h) - is the timed one; compressed cIRM against the oracle (bound 1e-4) and the enhanced waveform
    relative to its peak (bound 2e-3: the decompression slope reaches 100 near |m| = 9.9)."""
    from fsn_synthetic import make_noisy
    noisy2, ref, crm_ref = oracle_outputs
    x = make_noisy(max(batch, 2), length, seed=1234)
    x[0], x[-1] = noisy2[0], noisy2[1]
    enh, crm = model.enhance(torch.from_numpy(x).to(device), return_crm=True)
    torch.cuda.synchronize()
    rows = [0, x.shape[0] - 1]
    stft_ulp = stft_parity(noisy2, device)
    d_crm = float(np.abs(crm[rows].cpu().numpy() - crm_ref).max())
    d_enh = float(np.abs(enh[rows].cpu().numpy() - ref).max() / np.abs(ref).max())
    return {"max_abs_err_cirm_vs_oracle": d_crm, "bound_cirm": 1e-4, "mean_abs_err_cirm_vs_oracle":
            float(np.abs(crm[rows].cpu().numpy() - crm_ref).mean()), "max_rel_err_enhanced_vs_oracle": d_enh,
            "bound_enhanced": 2e-3, "cirm_range": [round(float(crm_ref.min()), 2), round(float(crm_ref.max()), 2)],
            "utterances": rows, "of_batch": int(x.shape[0]), "within_bound": bool(d_crm <= 1e-4 and d_enh <= 2e-3),
            "stft": stft_ulp,
            "note": "gate non-linearities are v_exp_f32 / v_rcp_f32 forms by choice (SURVEY 7 advises libm): this is "
                    "the margin they leave"}


def stft_parity(noisy2, device):
    """north_star: "STFT bins bit-pattern within 2 ULP".  Measured here on the checker's two utterances, in ULP of each
    frame's largest component (own-scale ULPs of cancellation bins are unbounded for ANY two FFTs): against the
    exactly rounded transform (the oracle's fp64 DFT of the fp32 frame x window product: bound 1) and against
    torch.stft on this box's CPU (MKL: the reference's own arithmetic).  The stated deviation lives here, in the line:
    MKL's fp32 FFT is itself up to ~3 ULP from the exact transform on long inputs (2.95 measured), so
    against MKL the tests allow 2 ULP on the short goldens and 3 on the 376-frame one."""
    import fullsubnet_amd
    from oracle import fullsubnet_oracle as O
    _, _, re, im = fullsubnet_amd.stft(torch.from_numpy(noisy2).to(device), N_FFT, HOP, N_FFT, return_phase=False)
    re, im = re.cpu().numpy(), im.cpu().numpy()
    win = torch.hann_window(N_FFT)
    _, _, ore, oim = O.stft(noisy2, window=win.numpy())
    mkl = torch.stft(torch.from_numpy(noisy2), N_FFT, HOP, N_FFT, window=win, return_complex=True)
    mre, mim = mkl.real.numpy(), mkl.imag.numpy()

    def ulps(ar, ai, br, bi):
        fmax = np.maximum(np.abs(br), np.abs(bi)).max(axis=1, keepdims=True)
        return float((np.maximum(np.abs(ar - br), np.abs(ai - bi)) / np.spacing(fmax.astype(np.float32))).max())

    return {"max_ulp_vs_exact_transform": round(ulps(re, im, ore, oim), 3), "bound_vs_exact": 1.0,
            "max_ulp_vs_torch_stft_cpu": round(ulps(re, im, mre, mim), 3),
            "torch_stft_cpu_vs_exact": round(ulps(mre, mim, ore, oim), 3),
            "bound_vs_torch_stft": {"north_star": 2.0, "tests": "2 on the short goldens, 3 on the 376-frame golden "
                                                                "(stated deviation: MKL itself is up to ~3 from exact)"},
            "scale": "ULP of each frame's largest component", "frames": int(re.shape[0] * re.shape[2])}


def training_parity(device, arith, saves=None):
    """The checker leg of the training figures: ONE step of a fresh model at exactly the timed shape on the inputs of
    tests/golden/fsn_train_c3.npz - the REFERENCE's own step (fullsubnet/trainer.py:41-71, use_amp = false, made by
    tests/golden/make_golden_train.py --config3) - loss, total gradient norm (what clip_grad_norm_ returns) and the
    worst parameter tensor's gradient norm, as relative errors.  Under the 16-bit arithmetic the same fp32 golden is the
    yardstick (tests/test_gpu_amp.py holds it to 3x the margins measured there)."""
    import ast
    import fullsubnet_amd
    from fullsubnet_amd.train import train_step
    from fsn_synthetic import make_noisy, make_params
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "fsn_train_c3.npz"))
    except OSError:
        return {"parity": None}
    meta = ast.literal_eval(str(z["meta"]))
    model = fullsubnet_amd.Model(num_freqs=F, look_ahead=LA, sequence_model="LSTM", fb_num_neighbors=0,
                                 sb_num_neighbors=NB, fb_output_activate_function="ReLU",
                                 sb_output_activate_function=False, fb_model_hidden_size=H_FB,
                                 sb_model_hidden_size=H_SB, norm_type="offline_laplace_norm",
                                 num_groups_in_drop_band=meta["groups"], weight_init=False)
    params = make_params(seed=meta["seed_w"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    model = model.to(device).train()
    model.train_arithmetic = arith
    if saves is not None:
        model.train_saves = saves
    scaler = torch.amp.GradScaler("cuda", enabled=arith != "f32")
    opt = fullsubnet_amd.ClipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
    noisy = torch.from_numpy(make_noisy(meta["batch"], meta["length"], seed=meta["seed_noisy"])).to(device)
    clean = torch.from_numpy((meta["clean_gain"] * make_noisy(meta["batch"], meta["length"], seed=meta["seed_clean"]))
                             .astype(np.float32)).to(device)
    loss = float(train_step(model, opt, noisy, clean, scaler=scaler))
    # ClipAdam leaves the unscaled, clipped gradients in .grad (what GradScaler.unscale_ + clip_grad_norm_ leave there)
    worst = max(abs(float(p.grad.norm()) - float(z["gnorm/" + k])) / (float(z["gnorm/" + k]) + 1e-30)
                for k, p in model.named_parameters())
    out = {"max_rel_err_vs_reference": round(max(abs(loss - float(z["loss"])) / float(z["loss"]),
                                                 abs(float(opt.total_norm) - float(z["total_norm"])) / float(z["total_norm"]),
                                                 worst), 9),
           "parity": {"against": "tests/golden/fsn_train_c3.npz: one step of the reference at this shape (fp32)",
                      "rel_err_loss": abs(loss - float(z["loss"])) / float(z["loss"]),
                      "rel_err_total_grad_norm": abs(float(opt.total_norm) - float(z["total_norm"])) / float(z["total_norm"]),
                      "worst_rel_err_tensor_grad_norm": worst}}
    del model, opt
    torch.cuda.empty_cache()
    return out


def family_figure(which, batch, peak_tflops, device):
    """Side figure for a sibling model (BASELINE configs 4 / 5): the whole path on `batch` x 3 s of synthetic audio,
    tools/bench_family.py:family_step - ms per step, frames/s and the fraction of the fp32-MFMA peak from SURVEY
    8(d)'s MFLOP per frame."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_family as BF
    pack = BF.build(which, device)
    m = BF.family_step(which, batch, device=device, steps=5, warmup=2, model_pack=pack)
    check = family_parity(BF, which, batch, pack, device)
    extra = {}
    if which == "fast":
        # BASELINE config 4 names a 1 -> 8 GPU scaling curve: the model has no cross-utterance term, so N ranks take batch / N
        # utterances each (one all-gather of the waveforms at the end, not in these).  PREDICTED from one rank's share measured
        # on this one GPU, like `strong_scaling_shares` of config 2.
        shares = {}
        for n in (2, 4, 8):
            ms = BF.family_step(which, batch // n, device=device, steps=5, warmup=2, model_pack=pack)["ms_per_step"]
            shares[str(n)] = {"utterances_per_rank": batch // n, "ms_per_step": round(ms, 3),
                              "predicted_speedup": round(m["ms_per_step"] / ms, 2)}
        extra["strong_scaling_shares"] = {**shares, "note": "PREDICTED: one rank's share of the batch at N ranks, each measured on "
                                          "this one GPU; excludes the all-gather of the enhanced waveforms"}
    del pack
    torch.cuda.empty_cache()
    return {**check, **extra, "ms_per_step": round(m["ms_per_step"], 3), "value": round(m["frames_per_s"], 1), "unit": "frames/s",
            "rtf_speedup_audio_s_per_s": round(m["rtf"], 1), "batch": batch, "samples": m["samples"],
            "sample_rate": m["sample_rate"], "frames_per_utterance": m["frames_per_utterance"],
            "mflop_per_frame": round(m["mflop_per_frame"], 1), "tflops": round(m["tflops"], 1),
            "frac_fp32_mfma_peak": round(m["tflops"] / peak_tflops, 3), "finite": m["finite"], "dtype": "f32"}


def family_parity(BF, which, batch, pack, device):
    """The checker leg of a side figure: the SAME batch the timed step ran (so the plan - kernels, tiles per workgroup -
    is the timed one), two of its utterances against the CPU oracle (oracle/model_family_oracle.py, pinned on the
    reference's goldens).  Fast FullSubNet: max |d| of the compressed mask (north-star bound 1e-4); Improved
    FullSubNet (waveform out, no compressed mask at its boundary): max |d| of the enhanced waveform relative to its
    peak.  The oracle is the checker only - nothing timed touches it."""
    from fsn_synthetic import make_fast_params, make_improved_params, make_noisy, IMPROVED_48K, IMPROVED_48K_769
    from oracle import fullsubnet_oracle as O
    from oracle import model_family_oracle as MF
    model, _, L, hop, sr, la = pack
    noisy_np = np.tile(make_noisy(min(batch, 8), L, seed=1), ((batch + 7) // 8, 1))[:batch]  # family_step's input
    rows = [0, batch - 1] if batch > 1 else [0]
    got = BF.parity_sample(which, model, torch.from_numpy(noisy_np).to(device), rows)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    t0 = time.perf_counter()
    if which == "fast":
        params = make_fast_params(seed=3)
        params["mel_scale.fb"] = model.mel_scale.fb.cpu().numpy()
        want = MF.fast_fullsubnet_forward(O.stft(noisy_np[rows])[0][:, None], params)
        err, scale, what = float(np.abs(got - want).max()), float(np.abs(want).max()), "compressed mask"
        out = {"max_abs_err_vs_oracle": err}
    elif which == "gru":
        # FullSubNet with sequence_model = "GRU" (fullsubnet/model.py:10-70; the oracle's GRU cell is pinned on the reference's
        # var_gru_b2 golden): the model's own random weights, two utterances (the offline norm takes its mean per utterance)
        params = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        want = O.fullsubnet_forward(O.stft(noisy_np[rows])[0][:, None], params, cell="GRU", num_groups_in_drop_band=1)
        err, scale, what = float(np.abs(got - want).max()), float(np.abs(want).max()), "compressed mask"
        out = {"max_abs_err_vs_oracle": err}
    else:
        cfg = {"improved48": IMPROVED_48K, "improved769": IMPROVED_48K_769}[which]
        want = MF.improved_fullsubnet_forward(noisy_np[rows], make_improved_params(cfg, seed=3), cfg,
                                              torch.hann_window(cfg["win_length"]).numpy()).reshape(len(rows), -1)
        scale, what = float(np.abs(want).max()), "enhanced waveform, relative to its peak"
        err = float(np.abs(got - want).max()) / scale
        out = {"max_abs_err_vs_oracle": err}
    out["parity"] = {"checked": what, "utterances": rows, "of_batch": batch, "output_scale": round(scale, 4),
                     "bound": 1e-4, "within_bound": bool(err <= 1e-4), "oracle_s": round(time.perf_counter() - t0, 1)}
    return out


def one_utterance_figure(model, length, device, reps=20):
    """Latency of ONE utterance through Model.enhance: eager, and as a hipGraph replay (fullsubnet_amd.GraphedCall; the
    replay must be bit-identical).  A side figure: the serving end of the path, where the step is a chain of small launches."""
    from fullsubnet_amd import GraphedCall
    from fsn_synthetic import make_noisy
    noisy = torch.from_numpy(make_noisy(1, length, seed=11)).to(device)
    graphed = GraphedCall(model.enhance)
    eager = model.enhance(noisy)
    same = bool(torch.equal(graphed(noisy), eager))
    res = {}
    for name, fn in (("eager", model.enhance), ("graph", graphed)):
        for _ in range(3):
            fn(noisy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(noisy)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps
    return {"samples": length, "eager_ms": round(1e3 * res["eager"], 3), "graph_replay_ms": round(1e3 * res["graph"], 3),
            "replay_bit_identical": same, "rtf_speedup_graph": round(length / 16000.0 / res["graph"], 1),
            "note": "ONE 3 s utterance, stft -> model -> mask -> istft: host-launched vs hipGraph replay of the same kernels"}


def measured_counters():
    """PMC figures of the dominant kernel REPLAYED from the committed rocprofv3 passes (profiles/rNN_pmc.json, produced
    by tools/rocprof_pmc.py from separate --pmc runs of `bench.py` at config 2): HBM bytes per launch (FETCH_SIZE x 2
    on gfx950 + WRITE_SIZE) and the MFMA-busy fraction of the launch.  Not measured in this run - the keys say so, and the
    stamp of the kernel sources the passes were taken on is compared with this tree's (`stale`: the sources have changed
    since; files of rounds 1 - 5 carry no stamp)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from rocprof_pmc import source_stamp
    here = source_stamp(ROOT)
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_hbm_traffic_end.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                j = json.load(f)
            d = j["dominant_kernel"]
            built = j.get("build") or {}
            stale = None if not built else built.get("csrc_sha256") != here["csrc_sha256"]
            return {"traffic": d.get("hbm_bytes_per_launch"), "mfma_busy": d.get("mfma_busy_frac"), "src": "profiles/" + name,
                    "taken_on": built or "unstamped (before round 6)", "stale": stale, "this_tree": here["csrc_sha256"]}
        except (OSError, KeyError, ValueError):
            continue
    return {"traffic": None, "mfma_busy": None, "src": None, "taken_on": None, "stale": None, "this_tree": here["csrc_sha256"]}


def hbm_stage_rooflines(stage_ms, b, length, T):
    """The two streaming stages against the HBM roofline (SURVEY 8(d)'s algorithmic bytes, fp32): stft = read 4 B L, write
    re / im / |X| 12 B F T; mask_istft (decompress + complex mask + irfft + overlap-add) = read mask 8 B F T + re / im 8 B F T,
    write 4 B L.  Durations: the stage's hipEvents on the launch stream."""
    out = {}
    for name, nbytes in (("stft", 4.0 * b * length + 12.0 * b * F * T), ("mask_istft", 16.0 * b * F * T + 4.0 * b * length)):
        ms = stage_ms.get(name, 0.0)
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[name] = {"bound": "hbm", "algorithmic_bytes": nbytes, "ms": round(ms, 4), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                     "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per step: whole job (strong) / per GPU (weak)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong (default, north star) = ONE --batch-utterance step sharded over the ranks; "
                         "weak = --batch utterances per rank.  The other mode is measured as a side figure.")
    ap.add_argument("--shard", choices=["auto", "utterances", "rows"], default="auto",
                    help="strong scaling: whole utterances per rank (all-gather of waveforms), or contiguous slices "
                         "of the batch x frequency rows of the sub-band model (all-gather of the full-band mask; "
                         "balances any batch over any number of ranks).  auto: utterances when they divide evenly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the CPU baseline with every host thread (minutes on a 256-thread box)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements (other scaling mode, opt-in arithmetic, training step)")
    ap.add_argument("--persistent", choices=["auto", "never"], default="auto",
                    help="never: take the kernels that run a whole recurrence as one launch out of every plan "
                         "(fsn_set_persistent_mode) - for ranks that SHARE a GPU (the gloo control-flow tests on a "
                         "one-GPU box), the one situation the residency contract excludes (include/fsn_hip.h)")
    ap.add_argument("--host-io", action="store_true",
                    help="side measurement for DESIGN.md: every step also copies its input from pinned host memory and "
                         "its result back (the PCIe-inclusive rate; never the headline `value`, which is HBM-resident)")
    ap.add_argument("--lib", default=None,
                    help="diagnosis: load this build of libfsn_hip.so (tools/build_variant.py) instead of the shipped one; the line says so")
    args = ap.parse_args()
    if args.lib:
        import fullsubnet_amd
        fullsubnet_amd._lib.LIB_PATH = os.path.abspath(args.lib)

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
    from fullsubnet_amd.parallel import enhance_row_sharded, shard_bounds
    from fsn_synthetic import make_noisy  # seeded synthetic input

    length = int(round(args.seconds * SR))
    T = 1 + length // HOP
    Tp = T + LA
    model, params = build_model(device)
    L = _lib.lib()
    _lib.set_persistent_mode(args.persistent)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_mode(scaling):
        """(step(), b_total, b_loc, rows_loc, description) of one scaling mode; inputs resident in HBM."""
        row_sharded = False
        if scaling == "weak" or world == 1:
            b_total, b_loc = args.batch * world, args.batch
            noisy_np = make_noisy(b_loc, length, seed=1234 + rank)
        else:
            b_total = args.batch
            row_sharded = args.shard == "rows" or (args.shard == "auto" and b_total % world != 0)
            if row_sharded:  # every rank holds the batch; its share is a slice of the B F sub-band rows
                b_loc = b_total
                noisy_np = make_noisy(b_total, length, seed=1234)
            else:
                lo, hi = shard_bounds(b_total, rank, world)
                b_loc = hi - lo
                noisy_np = make_noisy(b_total, length, seed=1234)[lo:hi]
        b_max = shard_bounds(b_total, 0, world)[1]
        noisy = torch.from_numpy(noisy_np).to(device)  # resident in HBM before the timed region
        gathered = send = None
        ag = {"ev": [], "ms": 0.0, "n": 0}  # the collective's own time: events on the launch stream around it
        if world > 1 and not row_sharded:
            gathered = torch.empty((world * b_max, length), dtype=torch.float32, device=device)
            send = torch.zeros((b_max, length), dtype=torch.float32, device=device)
        host_in = host_out = None
        if args.host_io:
            host_in = torch.from_numpy(noisy_np).pin_memory()
            host_out = torch.empty_like(host_in).pin_memory()

        def step():
            if args.host_io:
                noisy.copy_(host_in, non_blocking=True)
            if row_sharded:  # stft -> this rank's rows of the model -> all-gather of the mask -> decompress, apply, istft
                return enhance_row_sharded(model, noisy, N_FFT, HOP)
            enh = model.enhance(noisy, n_fft=N_FFT, hop_length=HOP)
            if args.host_io:
                host_out.copy_(enh, non_blocking=True)
            if world > 1:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if b_loc == b_max:
                    dist.all_gather_into_tensor(gathered, enh)  # RCCL over xGMI: re-assemble the node batch
                else:
                    send[:b_loc].copy_(enh)
                    dist.all_gather_into_tensor(gathered, send)
                e1.record()
                ag["ev"].append((e0, e1))
                return gathered
            return enh

        def all_gather_ms():
            """ms per step the all-gather took on this rank (read after a fence; includes waiting for the slowest peer)."""
            for e0, e1 in ag["ev"]:
                ag["ms"] += e0.elapsed_time(e1)
                ag["n"] += 1
            ag["ev"].clear()
            return ag["ms"] / max(ag["n"], 1)

        def reset_all_gather():
            ag["ev"].clear()
            ag["ms"], ag["n"] = 0.0, 0

        rows_loc = shard_bounds(b_total * F, 0, world)[1] if row_sharded else b_loc * F
        par = (f"row-shard x{world} ({rows_loc} of {b_total * F} sub-band rows per rank) + all-gather of the mask"
               if row_sharded else f"utterance-shard x{world}" + (" + all-gather of the waveforms" if world > 1 else ""))
        step.all_gather_ms, step.reset_all_gather, step.row_sharded = all_gather_ms, reset_all_gather, row_sharded
        return step, b_total, b_loc, rows_loc, par

    def run_mode(scaling, steps, warmup, profile):
        step, b_total, b_loc, rows_loc, par = make_mode(scaling)
        for _ in range(warmup):
            step()
        fence()
        step.reset_all_gather()
        if profile:
            _lib.profile_enable(True, device)  # hipEvents on the launch stream around every stage (no host sync inside)
        dt, stage_ms = timed_steps(step, fence, steps, (lambda: _lib.profile_read(device)) if profile else None)
        if profile:
            _lib.profile_enable(False, device)
        # what this rank's share runs on: the plan of its B utterances (row shard: of the whole batch it holds)
        plan = _lib.core_plan(model._cfg, max(1, -(-rows_loc // F)) if step.row_sharded else b_loc, T)
        per_rank = None
        if world > 1:
            # every rank's own clock, its all-gather time and its plan, for rank 0's line: a slow rank, a slow collective
            # and an unexpected plan are told apart at a glance the day an 8-GPU node runs this
            mine = torch.tensor([dt, step.all_gather_ms(), float(b_loc), float(plan["row_tiles_per_workgroup"]),
                                 float(plan["persistent_workgroups"]), float(plan["left_over_tiles"]),
                                 float(plan["group_clusters"]), float(plan["chunks"])], dtype=torch.float64, device=device)
            everyone = torch.empty((world * mine.numel(),), dtype=torch.float64, device=device)  # (flat: gloo insists)
            dist.all_gather_into_tensor(everyone, mine)
            everyone = everyone.view(world, mine.numel())
            per_rank = [{"rank": r, "ms_per_step": round(1e3 * v[0] / steps, 3), "all_gather_ms": round(v[1], 3),
                         "utterances": int(v[2]), "plan": {"row_tiles_per_workgroup": int(v[3]),
                                                           "persistent_workgroups": int(v[4]),
                                                           "left_over_tiles": int(v[5]), "group_clusters": int(v[6]),
                                                           "chunks": int(v[7])}}
                        for r, v in enumerate(everyone.cpu().tolist())]
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dict(dt=dt, stage_ms={k: v / steps for k, v in stage_ms.items()}, b_total=b_total, b_loc=b_loc,
                    rows_loc=rows_loc, par=par, steps=steps, plan=plan, per_rank=per_rank)

    scaling = args.scaling if world > 1 else "weak"  # one GPU: the two modes are the same workload
    head = run_mode(scaling, args.steps, args.warmup, profile=True)
    scaling_label = args.scaling  # at N = 1 both modes are the same 64-utterance step

    extras = {}
    if not args.no_extras and world > 1:  # the other mode, a few steps, for the record
        other = "weak" if scaling == "strong" else "strong"
        o = run_mode(other, max(2, min(args.steps, 5)), 1, profile=False)
        extras["other_scaling"] = {"scaling": other, "value": round(o["b_total"] * T * o["steps"] / o["dt"], 1),
                                   "unit": "frames/s", "ms_per_step": round(1e3 * o["dt"] / o["steps"], 3),
                                   "batch_total": o["b_total"], "parallelism": o["par"]}

    if rank == 0:
        dt, stage_ms, b_total, b_loc, rows_loc = head["dt"], head["stage_ms"], head["b_total"], head["b_loc"], head["rows_loc"]
        ms_per_step = 1e3 * dt / args.steps
        value = b_total * T * args.steps / dt
        # dominant kernel: the persistent sub-band recurrent kernel, launched twice per step (layers 0 and 1)
        # ... counted on the rows those two launches actually process (config 2: 256 workgroups x 64 rows = 16 384 of the
        # 16 448; the 64 left-over rows run as per-step launches beside them and are not in `ms` either)
        plan = head["plan"]
        rows_rec = plan["persistent_rows_all_chunks"]  # summed over the chunks (a remainder chunk has its own plan)
        rows_rec = min(rows_rec, rows_loc) if rows_rec > 0 else rows_loc
        rows_steps = float(rows_rec) * Tp
        rec_flops = 2.0 * (MAC_REC_L0 + MAC_REC_L1) * rows_steps  # both launches
        rec_ms = stage_ms.get("sb_rec_l0", 0.0) + stage_ms.get("sb_rec_l1", 0.0)
        achieved = rec_flops / (rec_ms * 1e-3) / 1e12 if rec_ms > 0 else 0.0
        path_flops = 2.0 * (MAC_FB * b_loc + MAC_SB_PER_BIN * rows_loc) * Tp
        at_config2 = world == 1 and b_loc == 64 and length == 48000
        pmc = measured_counters() if at_config2 else {"traffic": None, "mfma_busy": None, "src": None, "taken_on": None, "stale": None}
        traffic, mfma_busy, pmc_src = pmc["traffic"], pmc["mfma_busy"], pmc["src"]
        out = {
            "metric": "frames/sec (16 kHz, 512-FFT, hop 256), whole job", "value": round(value, 1),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling_label,
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" + (" (host buffers in and out over PCIe every step: --host-io)" if args.host_io else ""),
            "config": {"workload": f"FullSubNet inference (full_band_crm_mask path), 16 kHz, n_fft 512, hop 256, "
                                   f"n_neighbour 15, look_ahead 2, {b_total} x {args.seconds:g} s utterances per step "
                                   f"({b_loc} per GPU), offline_laplace_norm, full 257-bin mask per utterance "
                                   f"(BASELINE config 2" + (")" if b_total == 64 else f" at batch {b_total})"),
                       "batch_per_gpu": b_loc, "batch_total": b_total, "samples": length,
                       "frames_per_utterance": T, "parallelism": head["par"]},
            "rtf_speedup_audio_s_per_s": round(value / (SR / HOP), 1),
            "rtf_classic": round((SR / HOP) / value, 6),
            "roofline": {"bound": "mfma", "kernel": "lstm_rec_in_kernel<384,4,2> + lstm_rec_x_kernel<384,4,2> (the two "
                                                     "persistent sub-band recurrent launches of a step)",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         # PMC passes are separate rocprofv3 runs at config 2 (B = 64, 1 GPU)
                         "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc)",
                         "traffic_source": (f"replayed from {pmc_src} (separate rocprofv3 --pmc passes of this command "
                                            f"at config 2), NOT measured in this run") if pmc_src else None,
                         "mfma_busy_frac_replayed": mfma_busy, "pmc_source": pmc_src, "pmc_taken_on": pmc["taken_on"],
                         "pmc_stale": pmc["stale"],
                         "pmc_warning": ("the kernel sources have changed since the PMC passes were taken: re-run "
                                         "tools/gpu_run_full.sh") if pmc["stale"] else None,
                         "flops_per_launch": rec_flops / 2, "ms_per_launch": round(rec_ms / 2, 3),
                         "rows": {"persistent_pair": int(rows_rec), "of": int(rows_loc), "plan": plan},
                         "launches": {"sb_rec_l0": {"flops": 2.0 * MAC_REC_L0 * rows_steps,
                                                    "ms": round(stage_ms.get("sb_rec_l0", 0.0), 3)},
                                      "sb_rec_l1": {"flops": 2.0 * MAC_REC_L1 * rows_steps,
                                                    "ms": round(stage_ms.get("sb_rec_l1", 0.0), 3)}},
                         "whole_path_frac": round(path_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "roofline_hbm": hbm_stage_rooflines(stage_ms, b_loc, length, T),
        }
        if args.lib:
            out["library"] = f"DIAGNOSIS BUILD {args.lib} (not the shipped libfsn_hip.so)"
        if head["per_rank"] is not None:
            out["per_rank"] = head["per_rank"]
        out.update(extras)
    if not args.no_extras and world == 1 and args.batch % 8 == 0:
        # One rank's share of the strong-scaled batch at 2 / 4 / 8 GPUs (batch / N utterances), measured on this GPU:
        # the north-star target (>= 6x at 8 GPUs) is this time against ms_per_step, before the all-gather of the
        # waveforms (12 MB over xGMI) - a driver-visible PREDICTED curve for boxes without an 8-GPU node.
        keep = args.batch
        shares = {}
        for n in (2, 4, 8):
            args.batch = keep // n
            sh = run_mode("weak", max(5, args.steps), 3, profile=True)
            sh_ms = 1e3 * sh["dt"] / sh["steps"]
            shares[str(n)] = {"ranks": n, "utterances_per_rank": keep // n, "ms_per_step": round(sh_ms, 3),
                              "predicted_speedup": round(ms_per_step / sh_ms, 2),
                              "predicted_efficiency": round(ms_per_step / sh_ms / n, 3),
                              "stage_ms": {k: round(v, 3) for k, v in sh["stage_ms"].items()}}
        args.batch = keep
        out["strong_scaling_share"] = {
            **shares["8"], "predicted_speedup_at_8_gpus": shares["8"]["predicted_speedup"],
            "note": "one rank's share measured on ONE GPU (sub-band model on lstm2_group_kernel); excludes the all-gather"}
        out["strong_scaling_shares"] = {
            **shares, "note": "PREDICTED strong-scaling curve: one rank's share of the 64-utterance step at N ranks, each "
                              "measured on this one GPU; excludes the all-gather (12 MB of waveforms per node)"}
    if not args.no_extras and world == 1:
        # the opt-in split-precision kernels (Model.arithmetic -> cfg.arith), reported NEXT TO `value`, never as it.  Round 4:
        # the arithmetic passed its promotion criterion - error against the fp64 oracle at most 2x the fp32 path's on
        # adversarial inputs (tests/test_gpu_parity.py::test_f16x3_promotion_criterion, profiles/r04_f16x3_promotion.txt:
        # max-error ratios 1.26 - 1.86, rms 1.00 - 1.06) - and is a supported, still opt-in, inference arithmetic
        model.arithmetic = "f16x3"
        x = run_mode("weak", max(2, min(args.steps, 5)), 1, profile=True)
        model.arithmetic = "f32"
        out["split_f16x3"] = {
            "switch": 'Model.arithmetic = "f16x3"', "value": round(x["b_total"] * T * x["steps"] / x["dt"], 1),
            "unit": "frames/s", "ms_per_step": round(1e3 * x["dt"] / x["steps"], 3),
            "stage_ms": {k: round(v, 3) for k, v in x["stage_ms"].items()},
            "promotion": "error vs the fp64 oracle <= 2x the fp32 path's (max and rms) on noisy / tone-burst / 4000-step inputs: "
                         "tests/test_gpu_parity.py::test_f16x3_promotion_criterion",
            "note": "opt-in (fp32 operands split into two fp16 halves, three 16-bit MFMAs per product block, fp32 "
                    "accumulation); NOT the arithmetic of `value`"}
        for key, arith, saves in (("train_step", "f32", None), ("train_step_amp", "f16", None), ("train_step_amp_saves32", "f16", "32")):
            try:
                out[key] = training_step_ms(device, arith=arith, saves=saves)
            except Exception as e:  # a side figure must never break the benchmark line
                out[key] = {"error": str(e)[:200]}
        try:
            out["train_step_amp_shipped_batches"] = training_step_shipped_batches(device)
        except Exception as e:
            out["train_step_amp_shipped_batches"] = {"error": str(e)[:200]}
        # BASELINE configs 4 and 5 (fast_fullsubnet/model.py:143-202, improved_fullsubnet/model.py:541-591)
        # + FullSubNet with sequence_model = "GRU" at config 2's shape (a constructor option of fullsubnet/model.py:10-70 no shipped
        # TOML selects): its sub-band rows on the persistent many-row kernels with the GRU as a four-gate cell
        for key, which, b in (("fast_b256", "fast", 256), ("improved48_b32", "improved48", 32), ("gru_b64", "gru", 64)):
            try:
                out[key] = family_figure(which, b, PEAK_FP32_MFMA_TFLOPS, device)
            except Exception as e:
                out[key] = {"error": str(e)[:200]}
        # the sibling recipe that ships a training TOML with use_amp = true (fast_fullsubnet/train_shrinkSize2.toml:5,52: batch 72):
        # its bottleneck on the 16-bit persistent training kernels in pieces of whole clusters
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_family_train as BFT
            m = BFT.family_train_step("fast", 72, "f16", device=device)
            out["fast_train_b72_amp"] = {
                "ms_per_step": round(m["ms_per_step"], 2), "dtype": "f16 operands on the bottleneck (90 % of the products), fp32 elsewhere",
                "config": "fast_fullsubnet/train_shrinkSize2.toml: 72 x 49152 samples, cIRM MSE + clip_grad_norm_(10) + Adam, GradScaler",
                "loss": round(m["loss"], 6), "tflops": round(m["tflops"], 1), "skipped_steps": m["skipped_steps"],
                "parity": "tests/test_gpu_family.py::test_fast_fullsubnet_amp_step_vs_the_references_own_fp16_autocast_step"}
            torch.cuda.empty_cache()
        except Exception as e:
            out["fast_train_b72_amp"] = {"error": str(e)[:200]}
        # the launch-bound regime: ONE utterance, eager against a hipGraph replay of the whole call (fullsubnet_amd.GraphedCall)
        try:
            out["one_utterance"] = one_utterance_figure(model, length, device)
        except Exception as e:
            out["one_utterance"] = {"error": str(e)[:200]}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            torch.cuda.synchronize()
            out["cpu_baseline"] = cpu_baseline(params, length, args.cpu_budget, args.cpu_all_cores)
            out["parity"] = path_parity(model, out["cpu_baseline"].pop("_oracle_outputs"), args.batch, length, device)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None  # reported at N = 1 only
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
