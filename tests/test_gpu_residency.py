"""The residency contract of the persistent kernels (include/fsn_hip.h): fb_chain_kernel,
lstm2_group_kernel, lstm2_group_bptt_kernel and fb_chain_bptt_kernel need their whole grid resident, and RCCL's
kernels (the all-gather of a sharded batch, DDP's bucketed all-reduce during backward - base_trainer.py:32,
recipes/dns_interspeech_2020/train.py:29) are foreign kernels that may hold CUs beside them.  Here: a foreign "hog"
kernel of growing size beside config 2 at 8 utterances and beside a config-3 training step (bit-equal results, clean
status); a hog the persistent grid cannot outwait (the time-out surfaces as FsnTimeout, outputs NaN, never garbage;
the optimizer skips the update); the persistent kernels switched off; and RCCL itself at world size 1 around the same
stream graph.  Needs an MI355X:  python -m pytest tests -m gpu"""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import fullsubnet_oracle as O

pytestmark = pytest.mark.gpu

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


def make_model(fsn, seed=0, groups=1, **kw):
    params = O.make_params(seed=seed, **kw)
    m = fsn.Model(norm_type="offline_laplace_norm", num_groups_in_drop_band=groups, **MODEL_KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.cuda()


def wav(batch, length, seed):
    return torch.from_numpy(O.make_noisy(batch, length, seed=seed)).cuda()


class Hog:
    """A foreign kernel on its own stream: `workgroups` x 256 threads holding `lds` bytes of LDS each (and ~200
    registers per lane when heavy) for `ms` milliseconds (fsn_debug_hog)."""

    def __init__(self, fsn):
        self.fsn = fsn
        self.sink = torch.zeros(1, device="cuda")
        # HIP multiplexes streams onto a few hardware queues (round robin): a new stream may share the queue of the
        # current stream, and kernels of one queue never overlap.  Take a stream that demonstrably runs BESIDE the
        # current one: a 4 ms hog on it, then a trivial kernel on the current stream that must finish long before.
        self.stream, keep = None, []
        for _ in range(16):
            cand = torch.cuda.Stream()
            keep.append(cand)  # keep the rejected ones alive so that the next candidate lands on another queue
            self.stream = cand
            torch.cuda.synchronize()
            h0, h1, _ = self.launch(8, 1024, False, 4.0)
            e = torch.cuda.Event(enable_timing=True)
            self.sink.add_(0.0)
            e.record()
            torch.cuda.synchronize()
            if h0.elapsed_time(e) < 2.0:
                break
            self.stream = None
        assert self.stream is not None, "no stream that runs concurrently with the current one"

    def launch(self, workgroups, lds, heavy, ms):
        """Returns (start, end) events on the hog's stream and the host time the launch call took."""
        import time
        L = self.fsn._lib.lib()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            a.record()
            t0 = time.perf_counter()
            self.fsn._lib.check(L.fsn_debug_hog(workgroups, lds, 1 if heavy else 0, float(ms),
                                                self.fsn._lib.dev_ptr(self.sink), self.stream.cuda_stream))
            host_ms = 1e3 * (time.perf_counter() - t0)
            b.record()
        return a, b, host_ms

    def wait(self):
        self.stream.synchronize()


def timed(fn, with_events=False):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    b.synchronize()
    return (out, a.elapsed_time(b), a, b) if with_events else (out, a.elapsed_time(b))


# (workgroups, LDS bytes, heavy): a few workgroups .. one per CU .. two per CU; LDS-light / LDS-heavy / register-heavy
HOGS = [(8, 1024, True), (64, 64 * 1024, False), (128, 160 * 1024, False), (256, 64 * 1024, True),
        (256, 160 * 1024, False), (512, 32 * 1024, True)]


def test_inference_beside_a_foreign_kernel(fsn):
    """Config 2 at 8 utterances (the per-rank share at 8 GPUs: fb_chain_kernel on 256 workgroups, then lstm2_group_kernel
    on 512) while a foreign kernel holds a part of the chip or all of it: the workgroups that find no room start when
    it ends, the resident ones wait for them by the clock.  Bit-equal to the undisturbed run, status clean."""
    model = make_model(fsn, gain=2.0, mask_gain=24.0).eval()
    x = wav(8, 24000, 77)
    ref, t_ref = timed(lambda: model.enhance(x, return_crm=True))
    ref, t_ref = timed(lambda: model.enhance(x, return_crm=True))
    assert bool(torch.isfinite(ref[1]).all())
    hog = Hog(fsn)
    hog_ms = 60.0
    for wgs, lds, heavy in HOGS:
        torch.cuda.synchronize()
        h0, h1, host_ms = hog.launch(wgs, lds, heavy, hog_ms)
        got, t, e0, e1 = timed(lambda: model.enhance(x, return_crm=True), with_events=True)
        hog.wait()
        status, events = fsn._lib.stream_status(x.device, synchronize=True)
        # the timeline on one clock: the hog [0, h] and the call [s, e] relative to the hog's start
        h, s, e = h0.elapsed_time(h1), h0.elapsed_time(e0), h0.elapsed_time(e1)
        print(f"hog {wgs:4d} wgs x {lds // 1024:3d} KB {'heavy' if heavy else 'light'}: enhance {t:6.1f} ms "
              f"(undisturbed {t_ref:.1f} ms); hog ran [0, {h:.1f}] ms, the call [{s:.1f}, {e:.1f}] ms, "
              f"launching the hog took {host_ms:.2f} ms of host time")
        assert (status, events) == (0, 0)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (wgs, lds, heavy)
        assert h >= 0.9 * hog_ms and s < 0.5 * hog_ms, "the hog was not running when the call started: vacuous"
        if wgs >= 256 and lds >= 160 * 1024:
            # a hog that owns every CU's LDS: nothing of the path can finish before it ends
            assert e >= 0.9 * h, (e, h)


def test_training_step_beside_a_foreign_kernel(fsn):
    """One training step at config 3's per-rank shape (16 x 49 152 samples, drop_band groups 2: the full-band chain and
    its BPTT, the group kernel with saves and the group BPTT kernel - all four persistent kernels) while foreign
    kernels come and go on another stream, as DDP's bucketed all-reduces do during backward: loss, gradients and updated
    parameters bit-equal to the undisturbed step."""
    from fullsubnet_amd.train import train_step
    noisy, clean = wav(16, 49152, 5), 0.7 * wav(16, 49152, 6)

    def run(hog=None):
        model = make_model(fsn, seed=3, groups=2).train()
        opt = fsn.ClipAdam(model.parameters(), lr=1e-3)
        marks = []
        if hog is not None:
            torch.cuda.synchronize()
            for wgs, lds, heavy in [(64, 64 * 1024, True), (256, 160 * 1024, False), (128, 96 * 1024, True)]:
                marks.append(hog.launch(wgs, lds, heavy, 25.0))  # back to back on the hog's stream: ~75 ms beside a ~45 ms step
        (loss, t, e0, e1) = timed(lambda: train_step(model, opt, noisy, clean), with_events=True)
        torch.cuda.synchronize()
        if marks:
            h0, h1 = marks[0][0], marks[-1][1]
            print(f"training step {t:.1f} ms; hogs ran [0, {h0.elapsed_time(h1):.1f}] ms, the step "
                  f"[{h0.elapsed_time(e0):.1f}, {h0.elapsed_time(e1):.1f}] ms")
            assert h0.elapsed_time(e0) < 25.0, "the hogs were not running when the step started: vacuous"
        else:
            print(f"training step undisturbed {t:.1f} ms")
        assert fsn._lib.stream_status(noisy.device) == (0, 0)
        assert opt.skipped_steps() == 0
        return loss.item(), [p.grad.clone() for p in model.parameters()], [p.detach().clone() for p in model.parameters()]

    ref = run()
    got = run(Hog(fsn))
    assert np.isfinite(ref[0]) and got[0] == ref[0]
    for a, b in zip(ref[1] + ref[2], got[1] + got[2]):
        assert torch.isfinite(b).all() and torch.equal(a, b)


def test_a_wait_that_runs_out_is_reported_not_silent(fsn):
    """A foreign kernel that holds half the chip for longer than the wait bound (set to 5 ms here; 20 s by default):
    the resident half of fb_chain_kernel gives up, the launch ends with NaN outputs (never garbage, never a hang), the
    stream's sticky status is raised - fsn_stream_status reports it, every later persistent launch on the stream is
    refused with FSN_ERR_TIMEOUT until the record is cleared - and afterwards the stream works as before."""
    model = make_model(fsn, gain=2.0, mask_gain=24.0).eval()
    x = wav(8, 12000, 78)
    ref = model.enhance(x, return_crm=True)
    torch.cuda.synchronize()
    hog = Hog(fsn)
    fsn._lib.set_persistent_timeout_ms(5)
    try:
        hog.launch(128, 160 * 1024, False, 400.0)
        enh, crm = model.enhance(x, return_crm=True)
        with pytest.raises(fsn._lib.FsnTimeout):
            fsn._lib.stream_status(x.device, synchronize=True)
        status, events = fsn._lib.stream_status(x.device, raise_on_timeout=False)
        assert status != 0 and events >= 1
        assert bool(torch.isnan(crm).all()) and bool(torch.isnan(enh).all())
        with pytest.raises(fsn._lib.FsnTimeout):
            model.enhance(x)
        hog.wait()
    finally:
        fsn._lib.set_persistent_timeout_ms(20000)
        fsn._lib.stream_status_clear(x.device)
    assert fsn._lib.stream_status(x.device) == (0, 0)
    again = model.enhance(x, return_crm=True)
    assert torch.equal(again[0], ref[0]) and torch.equal(again[1], ref[1])


def test_a_poisoned_step_skips_the_update(fsn):
    """one poisoned launch must not turn the weights and Adam's moments into NaN for good.  A non-finite
    gradient norm skips the update on the device (no host sync), counts it, and does not advance Adam's step count:
    the next clean step is bit-equal to the step an undisturbed optimizer takes."""
    from fullsubnet_amd.train import train_step
    noisy, clean = wav(4, 2560, 11), 0.7 * wav(4, 2560, 12)

    def fresh():
        model = make_model(fsn, seed=3, groups=2).train()
        return model, fsn.ClipAdam(model.parameters(), lr=1e-3, clip_grad_norm_value=10.0)

    m_ref, o_ref = fresh()
    loss_ref = train_step(m_ref, o_ref, noisy, clean).item()

    model, opt = fresh()
    before = [p.detach().clone() for p in model.parameters()]
    bad = noisy.clone()
    bad[0, 100] = float("nan")  # a NaN in the input: NaN loss, NaN gradients, like a poisoned launch
    loss_bad = train_step(model, opt, bad, clean)
    assert not np.isfinite(loss_bad.item())
    assert opt.skipped_steps() == 1
    for p, q in zip(model.parameters(), before):
        assert torch.equal(p.detach(), q)
    for st in opt.state.values():
        assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0
    assert all(int(st["step"]) == 0 for st in opt.state_dict()["state"].values())  # applied updates, not calls
    loss = train_step(model, opt, noisy, clean).item()
    assert loss == loss_ref
    for p, q in zip(model.parameters(), m_ref.parameters()):
        # the bias corrections of the shifted step count are formed on the device (fp64 pow): equal to the last bit or so
        assert (p.detach() - q.detach()).abs().max().item() <= 1e-9
    assert opt.skipped_steps() == 1


def test_persistent_kernels_switched_off(fsn):
    """FSN_PERSISTENT_NEVER: the same calls on the per-step paths (what a caller that cannot rule out a second process
    on its GPU selects) - same results to rounding, inference and training."""
    from fullsubnet_amd.train import train_step
    model = make_model(fsn, gain=2.0, mask_gain=24.0).eval()
    x = wav(8, 12000, 79)
    noisy, clean = wav(16, 8192, 5), 0.7 * wav(16, 8192, 6)

    def step():
        m = make_model(fsn, seed=3, groups=2).train()
        loss = train_step(m, torch.optim.SGD(m.parameters(), lr=0.0), noisy, clean)
        return loss.item(), [p.grad.clone() for p in m.parameters()]

    ref = model.enhance(x, return_crm=True)
    ref_step = step()
    fsn._lib.set_persistent_mode("never")
    try:
        got = model.enhance(x, return_crm=True)
        got_step = step()
    finally:
        fsn._lib.set_persistent_mode("auto")
    assert (got[1] - ref[1]).abs().max().item() <= 5e-5
    assert (got[0] - ref[0]).abs().max().item() <= 1e-4 * ref[0].abs().max().item()
    assert abs(got_step[0] - ref_step[0]) <= 1e-5 * abs(ref_step[0])
    for a, b in zip(ref_step[1], got_step[1]):
        assert (a - b).abs().max().item() <= 2e-4 * max(a.abs().max().item(), 1e-6)
    back = model.enhance(x, return_crm=True)
    assert torch.equal(back[1], ref[1])
    assert fsn._lib.persist_stats()[2] == 0, "a persistent launch did not report its footprint to the gate"


# ---- RCCL itself (world size 1: one rank per GPU is all a one-GPU box allows) -----------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rccl_worker(rank, world, port):
    import torch.distributed as dist
    import fullsubnet_amd as fsn
    from fullsubnet_amd.parallel import enhance_row_sharded, enhance_sharded
    from fullsubnet_amd.train import train_step
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        model = make_model(fsn, gain=2.0, mask_gain=24.0).eval()
        x = wav(8, 12000, 80)
        fused = model.enhance(x)
        # the sharded entry points with their RCCL all-gathers (all_gather_into_tensor) on the real backend
        utt = enhance_sharded(model.enhance, x)
        rows = enhance_row_sharded(model, x)
        assert torch.equal(utt, fused)
        assert (rows - fused).abs().max().item() <= 1e-4 * fused.abs().max().item()
        # an all-gather in flight on RCCL's stream while the persistent kernels run on the compute stream
        big = torch.randn(32 << 20, device="cuda")
        out = torch.empty_like(big)
        work = dist.all_gather_into_tensor(out, big, async_op=True)
        again = model.enhance(x)
        work.wait()
        assert torch.equal(again, fused) and torch.equal(out, big)
        # ... and issued by the caller on a SIDE stream, several in a row, while the persistent kernels (full-band chain +
        # group kernel: 8 utterances x 190 steps, ~12 ms) are resident on the compute stream: RCCL's kernels vs the
        # residency contract (include/fsn_hip.h) - results bit-equal, the two streams really overlapped, no time-out record
        long_x = wav(8, 48000, 81)
        want = model.enhance(long_x)
        side, main = torch.cuda.Stream(), torch.cuda.current_stream()
        outs = [torch.empty_like(big) for _ in range(6)]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        torch.cuda.synchronize()
        before = fsn._lib.persist_stats()[0]
        ev[0].record(main)
        side.wait_stream(main)
        ev[3].record(main)
        got = model.enhance(long_x)  # enqueued: ~12 ms of device time ahead of the host
        ev[4].record(main)
        with torch.cuda.stream(side):  # the collectives enter while those kernels run
            ev[1].record(side)
            for o in outs:
                dist.all_gather_into_tensor(o, big)
            ev[2].record(side)
        main.wait_stream(side)
        torch.cuda.synchronize()
        assert fsn._lib.persist_stats()[0] - before >= 2, "the call must have run on the persistent launches"
        assert torch.equal(got, want) and all(torch.equal(o, big) for o in outs)
        t = [ev[0].elapsed_time(e) for e in ev]  # ms since ev[0]: collectives [t1, t2], model [t3, t4]
        print(f"RCCL all-gathers on a side stream {t[1]:.2f} .. {t[2]:.2f} ms, persistent kernels {t[3]:.2f} .. {t[4]:.2f} ms")
        assert t[1] < t[4] and t[3] < t[2], t  # the intervals intersect
        assert fsn._lib.stream_status(x.device) == (0, 0)
        # DistributedDataParallel around Model (base_trainer.py:32): bucketed all-reduce hooks fire during backward,
        # beside the persistent BPTT kernels; gradients = the plain model's
        noisy, clean = wav(16, 8192, 5), 0.7 * wav(16, 8192, 6)
        ref = make_model(fsn, seed=3, groups=2).train()
        train_step(ref, torch.optim.SGD(ref.parameters(), lr=0.0), noisy, clean)
        ddp = torch.nn.parallel.DistributedDataParallel(make_model(fsn, seed=3, groups=2).train(), device_ids=[0])
        loss = train_step(ddp, torch.optim.SGD(ddp.parameters(), lr=0.0), noisy, clean)
        assert torch.isfinite(loss)
        for (k, p), (_, q) in zip(ref.named_parameters(), ddp.module.named_parameters()):
            assert torch.equal(p.grad, q.grad), k
        assert fsn._lib.stream_status(x.device) == (0, 0)
    finally:
        dist.destroy_process_group()


def test_persistent_launches_from_several_streams_share_the_chip_when_provably_placeable(fsn):
    """The gate of fsn_api.hip on two-layer stacks of 20 - 30 rows (chain kernel with two row tiles: 192 workgroups, two
    resident per CU - the large band sections of improved_fullsubnet/model.py at batch 1).  Two of them on two streams
    are admitted side by side (no wait inserted, and they finish sooner than one after the other); a third on a third
    stream has to wait for the oldest (three would not be placeable in every dispatch order).  Results equal the
    one-stream run bit for bit, every launch reported its footprint, the streams' status stays clean."""
    from fullsubnet_amd.sequence_model import SequenceModel
    torch.manual_seed(5)
    T = 1000
    rows = (20, 25, 30)
    models = [SequenceModel(40 + 8 * i, 2, 384, 2, False, "LSTM", None).cuda() for i in range(3)]
    xs = [torch.randn(n, 40 + 8 * i, T, device="cuda") for i, n in enumerate(rows)]
    main = torch.cuda.current_stream()
    sink = torch.zeros(1, device="cuda")

    def overlap(a, b):
        """Do kernels of streams a and b run side by side?  (HIP multiplexes streams onto a few hardware queues; two
        streams of one queue never overlap - see Hog.)  Two 4 ms kernels of 8 workgroups: ~4 ms together, or ~8."""
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in (a, b):
            st.wait_stream(main)
            fsn._lib.check(fsn._lib.lib().fsn_debug_hog(8, 1024, 0, 4.0, fsn._lib.dev_ptr(sink), st.cuda_stream))
        for st in (a, b):
            main.wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) < 6.0

    pool = [torch.cuda.Stream() for _ in range(12)]
    pair = next(((a, b) for i, a in enumerate(pool) for b in pool[i + 1:] if overlap(a, b)), None)
    assert pair is not None, "no two streams that run concurrently"
    streams = [pair[0], pair[1], next(st for st in pool if st not in pair)]

    def on_one(n):
        with torch.no_grad():
            return [m(x) for m, x in zip(models[:n], xs[:n])]

    def on_streams(n):
        outs = []
        for st, m, x in zip(streams[:n], models[:n], xs[:n]):
            st.wait_stream(main)
            with torch.cuda.stream(st), torch.no_grad():
                outs.append(m(x))
        for st in streams[:n]:
            main.wait_stream(st)
        return outs

    ref = on_one(3)
    torch.cuda.synchronize()  # (those launches are retired from the gate's list when the next one looks)

    def counted(n):
        before = fsn._lib.persist_stats()
        # a small kernel that holds the CALLER's stream for 150 ms first: the side streams wait for it, so none of the
        # launches below can have finished (or started) by the time the last one is admitted - what the gate decides
        # does not depend on how fast the host enqueues
        fsn._lib.check(fsn._lib.lib().fsn_debug_hog(8, 1024, 0, 150.0, fsn._lib.dev_ptr(sink), main.cuda_stream))
        outs = on_streams(n)
        torch.cuda.synchronize()
        after = fsn._lib.persist_stats()
        assert after[2] == 0, "a persistent launch did not report its footprint to the gate"
        for o, r in zip(outs, ref):
            assert torch.equal(o, r)
        return after[0] - before[0], after[1] - before[1]

    two, three = counted(2), counted(3)
    print(f"gate: (launches, waits) two streams {two}, three streams {three}")
    assert two == (2, 0), "two chain launches of 192 workgroups at two per CU are placeable together"
    assert three[0] == 3 and three[1] >= 1, "a third one is not: it has to wait"
    for st in streams:
        with torch.cuda.stream(st):
            assert fsn._lib.stream_status(synchronize=True) == (0, 0)

    def timed(fn, n):
        for _ in range(2):
            fn(n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn(n)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 3

    t_one, t_two = timed(on_one, 2), timed(on_streams, 2)
    print(f"two chain stacks of {T} steps: one stream {t_one:.2f} ms, two streams {t_two:.2f} ms")
    assert t_two < 0.85 * t_one, (t_one, t_two)


def test_rccl_world_size_one(fsn):
    """`nccl` (= RCCL) has run on this stream graph at least once: process-group init on the GPU, the two sharded
    enhancement entry points, an all-gather in flight beside the persistent kernels, DDP around Model."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(1, _free_port()), nprocs=1, join=True)
