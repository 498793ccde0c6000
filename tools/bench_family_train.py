"""Time one training step of the sibling recipes that ship training TOMLs, on one MI355X:
python tools/bench_family_train.py [fast|fullband] [batch] [f32|f16|bf16]   (fast_fullsubnet/train_shrinkSize2.toml: batch 72 x
3.072 s, use_amp = true; fullband_baseline/train.toml).  The LSTM / Linear blocks run on the library's training entries, the glue
between them is autograd-tracked tensor algebra.  f16 / bf16: the trainer's autocast arithmetic + GradScaler -
Fast FullSubNet's bottleneck (two layers x 384 units on B x 64 rows) then runs on the 16-bit persistent training kernels.
`family_train_step(which, B, arith)` is also what bench.py's side figure `fast_train_b72_amp` calls."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd  # noqa: E402
from fullsubnet_amd.train import train_step  # noqa: E402
from fsn_synthetic import make_fast_params, make_fullband_params, make_noisy  # noqa: E402


def build(which, device="cuda"):
    """(model, MAC per utterance and computed frame) of a sibling recipe's training configuration."""
    if which == "fast":
        from fullsubnet_amd.fast_fullsubnet import Model
        model = Model(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
                      bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
                      encoder_output_num_neighbors=0, norm_type="offline_laplace_norm", weight_init=False)
        sd = {k: torch.from_numpy(v) for k, v in make_fast_params(seed=3).items()}
        sd["mel_scale.fb"] = model.mel_scale.fb.clone()
        # encoder + decoder at the frame rate, the bottleneck (64 bands) at half rate
        mac = 4 * 384 * (64 + 384) + 4 * 257 * (384 + 257) + 257 * 64 + 4 * 512 * (128 + 512) + 4 * 512 * 1024 + 512 * 514 \
            + 64 * (4 * 384 * (12 + 384) + 4 * 384 * 768 + 384) // 2
    else:
        from fullsubnet_amd.fullband_baseline import Model
        model = Model(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=False, look_ahead=2,
                      norm_type="offline_laplace_norm", weight_init=False)
        sd = {k: torch.from_numpy(v) for k, v in make_fullband_params(seed=3).items()}
        mac = 4 * 512 * (257 + 512) + 2 * 4 * 512 * 1024 + 512 * 514
    model.load_state_dict(sd, strict=True)
    return model.to(device).train(), mac


def family_train_step(which="fast", B=72, arith="f32", L=49152, steps=3, warmup=2, device="cuda"):
    model, mac = build(which, device)
    model.train_arithmetic = arith
    scaler = torch.amp.GradScaler("cuda", enabled=arith != "f32")
    opt = fullsubnet_amd.ClipAdam(model.parameters(), lr=1e-3)
    noisy = torch.from_numpy(make_noisy(B, L, seed=1)).to(device)
    clean = torch.from_numpy(0.7 * make_noisy(B, L, seed=2)).to(device)
    for _ in range(warmup):
        loss = train_step(model, opt, noisy, clean, scaler=scaler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(model, opt, noisy, clean, scaler=scaler)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    T = 1 + L // 256
    return {"ms_per_step": 1e3 * dt, "loss": loss.item(), "tflops": 3 * 2 * mac * B * (T + 2) / dt / 1e12,
            "frames_per_s": B * T / dt, "skipped_steps": opt.skipped_steps(), "scale": scaler.get_scale() if arith != "f32" else None,
            "peak_mem_gib": torch.cuda.max_memory_allocated() / 2**30}


if __name__ == "__main__":
    for a in sys.argv[1:]:
        if a.startswith("--lib="):  # a diagnosis build (tools/build_variant.py)
            fullsubnet_amd._lib.LIB_PATH = a[6:]
            sys.argv.remove(a)
    which = sys.argv[1] if len(sys.argv) > 1 else "fast"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 72
    arith = sys.argv[3] if len(sys.argv) > 3 else "f32"
    m = family_train_step(which, B, arith)
    print(f"train step {which} B={B} {arith}: {m['ms_per_step']:.1f} ms, loss {m['loss']:.5f}, ~{m['tflops']:.1f} TFLOP/s "
          f"({m['frames_per_s']:.0f} frames/s), peak mem {m['peak_mem_gib']:.1f} GiB")
