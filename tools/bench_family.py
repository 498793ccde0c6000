"""Time the sibling model families on one MI355X (BASELINE configs 1, 4 and 5):
python tools/bench_family.py [fast|fullband|gru|improved16|improved48|improved769] [batch] [units=r/w]
units=r/w (improved* only): time what rank r of w computes under the frequency-axis shard (its share of every
section's units; the all-gather is not part of this single-GPU measurement).
`family_step(which, B)` is also what bench.py's side figures `fast_b256` / `improved48_b32` call."""
import contextlib
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fullsubnet_amd import decompress_cIRM, istft, stft  # noqa: E402
from fsn_synthetic import (IMPROVED_16K, IMPROVED_48K, IMPROVED_48K_769, make_fast_params, make_fullband_params,  # noqa: E402
                           make_improved_params, make_noisy)

@contextlib.contextmanager
def collector_paused():
    """Timed regions of a few tens of milliseconds: a generation-2 pass of CPython's cyclic collector over torch's
    ~10^6 objects is a ~40 ms host pause that lands in one of them now and then (measured with the HIP API trace: one
    utterance of config 5, 4.7 ms per step, read 8.5 - 12.7 ms with one such pause in five steps).  Collected before,
    held during."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


DEFAULT_BATCH = {"fast": 256, "improved48": 32, "improved769": 32, "improved16": 32}


def build(which, device="cuda"):
    """(model, mac_per_frame_per_utterance, samples, hop, sample_rate, look_ahead) of one family at 3 s clips."""
    L, hop = 48000, 256
    if which.startswith("improved"):
        from fullsubnet_amd.improved_fullsubnet import Model
        cfg = {"improved48": IMPROVED_48K, "improved769": IMPROVED_48K_769}.get(which, IMPROVED_16K)
        L = 48000 if which == "improved16" else 144000  # 3 s
        hop = cfg["hop_length"]
        model = Model(**cfg)
        sd = {k: torch.from_numpy(v) for k, v in make_improved_params(cfg, seed=3).items()}
        F = cfg["num_freqs"] - 1
        mmac = 4 * 512 * (F + 512) + 4 * 512 * 1024 + 512 * F  # full-band model
        cuts = [0] + list(cfg["freq_cutoffs"]) + [F]
        for i, c in enumerate(cfg["sb_num_center_freqs"]):
            units = (cuts[i + 1] - cuts[i]) // c
            k_in = 2 * (c + 30)
            mmac += units * (4 * 384 * (k_in + 384) + 4 * 384 * 768 + 384 * 2 * c)
    elif which == "fast":
        from fullsubnet_amd.fast_fullsubnet import Model
        model = Model(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
                      bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
                      encoder_output_num_neighbors=0)
        sd = {k: torch.from_numpy(v) for k, v in make_fast_params(seed=3).items()}
        sd["mel_scale.fb"] = model.mel_scale.fb.clone()
        mmac = 62.9e6  # MAC / frame / utterance
    elif which == "gru":
        # FullSubNet with sequence_model = "GRU" (fullsubnet/model.py:10-70: a constructor option no shipped TOML selects): the
        # composed configuration - SequenceModel blocks on the GRU kernels, the glue between them on the library's glue kernels
        from fullsubnet_amd.model import Model
        torch.manual_seed(3)
        model = Model(num_freqs=257, look_ahead=2, sequence_model="GRU", fb_num_neighbors=0, sb_num_neighbors=15,
                      fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                      sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=True)
        sd = model.state_dict()
        mmac = 3 * 512 * (257 + 512) + 3 * 512 * 1024 + 512 * 257 + 257 * (3 * 384 * (32 + 384) + 3 * 384 * 768 + 384 * 2)
    else:
        from fullsubnet_amd.fullband_baseline import Model
        model = Model(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=None, look_ahead=2,
                      weight_init=False)
        sd = {k: torch.from_numpy(v) for k, v in make_fullband_params(seed=3).items()}
        mmac = 6.032384e6
    model.load_state_dict(sd, strict=True)
    la = 0 if which.startswith("improved") else 2
    sr = 48000 if which in ("improved48", "improved769") else 16000
    return model.to(device).eval(), mmac, L, hop, sr, la


def enhance_fn(which, model):
    @torch.no_grad()
    def enhance(y):
        if which.startswith("improved"):
            return model(y)  # waveform in, waveform out (improved_fullsubnet/model.py:541-591)
        mag, _, re, im = stft(y, 512, 256, 512, return_phase=False)
        crm = decompress_cIRM(model(mag.unsqueeze(1)).permute(0, 2, 3, 1))
        er = crm[..., 0] * re - crm[..., 1] * im
        ei = crm[..., 1] * re + crm[..., 0] * im
        return istft((er, ei), 512, 256, 512, length=y.size(-1), input_type="real_imag")
    return enhance


def parity_sample(which, model, noisy, rows):
    """What the parity checker compares for a family, computed on the WHOLE batch `noisy` (so the kernels are the ones
    the timed step runs) and returned for the utterances `rows` only: the compressed mask [n, 2, F, T] for the
    magnitude-in / mask-out models, the enhanced waveform [n, L] for Improved FullSubNet.  No oracle in here: bench.py
    holds the checker."""
    with torch.no_grad():
        if which.startswith("improved"):
            out = model(noisy)
            return out.reshape(noisy.shape[0], -1)[rows].cpu().numpy()
        mag = stft(noisy, 512, 256, 512, return_phase=False)[0]
        return model(mag.unsqueeze(1))[rows].cpu().numpy()


def family_step(which, B, device="cuda", steps=5, warmup=2, model_pack=None):
    """Whole path (stft -> model -> decompress -> mask -> istft; waveform to waveform for Improved FullSubNet) on B x 3 s
    of synthetic audio resident in HBM: dict(ms_per_step, frames_per_s, rtf, tflops, finite)."""
    model, mmac, L, hop, sr, la = model_pack or build(which, device)
    noisy = torch.from_numpy(make_noisy(min(B, 8), L, seed=1)).to(device).repeat((B + 7) // 8, 1)[:B].contiguous()
    enhance = enhance_fn(which, model)
    for _ in range(warmup):
        out = enhance(noisy)
    torch.cuda.synchronize()
    with collector_paused():
        t0 = time.perf_counter()
        for _ in range(steps):
            out = enhance(noisy)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    T = 1 + L // hop
    return {"ms_per_step": dt * 1e3, "frames_per_s": B * T / dt, "rtf": B * L / sr / dt,
            "tflops": 2 * mmac * B * (T + la) / dt / 1e12, "mflop_per_frame": 2 * mmac / 1e6, "batch": B,
            "frames_per_utterance": T, "samples": L, "sample_rate": sr, "finite": bool(torch.isfinite(out).all())}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fast"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else DEFAULT_BATCH.get(which, 1)
    pack = build(which)
    model = pack[0]
    shard = [a for a in sys.argv[3:] if a.startswith("units=")]
    which_label = ""
    if shard:
        r, w = (int(v) for v in shard[0][6:].split("/"))
        sb = model.sb_model

        def one_rank(noisy_mag, fb_output, unit_group=None):
            local = sb.forward_units(noisy_mag, fb_output, r, w)
            # stand-in for the gathered result (timing only): this rank's units repeated to the full count
            full = [t.repeat((n + max(t.shape[0], 1) - 1) // max(t.shape[0], 1), 1, 1, 1, 1)[:n] if t.shape[0]
                    else t.new_zeros((n,) + tuple(t.shape[1:]))
                    for t, n in zip(local, sb.num_units(noisy_mag.size(2)))]
            return sb.assemble_units(full)

        sb.forward = one_rank
        which_label = f" units {r}/{w}"
    m = family_step(which, B, model_pack=pack)
    print(f"{which}{which_label} B={B}: {m['ms_per_step']:.2f} ms / batch, {m['frames_per_s']:.0f} frames/s "
          f"({m['rtf']:.0f} x real time), ~{m['tflops']:.1f} TFLOP/s, finite={m['finite']}")


if __name__ == "__main__":
    main()
