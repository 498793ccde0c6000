"""Soak run of the training step (AMP and fp32, side streams on): several hundred steps on changing data; prints the loss
trajectory, the stream status, the persistent-launch statistics and the slowest step.  python tools/soak_train.py [batch=16]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fullsubnet_amd
from fullsubnet_amd.train import train_step
from fsn_synthetic import make_noisy, make_params
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for ARITH, K in (("f16", 400), ("f32", 150)):
    model = fullsubnet_amd.Model(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                                 fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                                 sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=2, weight_init=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_params(seed=3).items()})
    model = model.cuda().train(); model.train_arithmetic = ARITH
    scaler = torch.amp.GradScaler("cuda", enabled=ARITH != "f32")
    opt = fullsubnet_amd.ClipAdam(model.parameters(), lr=1e-4)
    losses, slowest = [], 0.0
    t0 = time.perf_counter()
    for i in range(K):
        ts = time.perf_counter()
        noisy = torch.from_numpy(make_noisy(B, 49152, seed=100 + i % 7)).cuda()
        clean = torch.from_numpy(0.7 * make_noisy(B, 49152, seed=200 + i % 7)).cuda()
        loss = train_step(model, opt, noisy, clean, scaler=scaler)
        if i % 50 == 0: losses.append(round(loss.item(), 5))
        if i % 10 == 9:
            torch.cuda.synchronize(); slowest = max(slowest, (time.perf_counter() - ts))
    torch.cuda.synchronize()
    st = fullsubnet_amd._lib.stream_status(synchronize=True)
    print(ARITH, K, "steps", f"{(time.perf_counter() - t0) / K * 1e3:.1f} ms/step (incl. host data)", "losses", losses, "status", st, f"slowest synchronised step {slowest * 1e3:.0f} ms",
          "persist", fullsubnet_amd._lib.persist_stats(), "finite", all(torch.isfinite(p).all().item() for p in model.parameters()))
