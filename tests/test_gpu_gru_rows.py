"""nn.GRU with MANY rows in inference - the sub-band model of a GRU FullSubNet (audio_zen/model/module/sequence_model.py:59-66
under fullsubnet/model.py:121-128) - on the LSTM's persistent many-row kernels with the GRU written as a four-gate cell
(FSN_REC_GRU in lstm_kernels.hip, fsn_gru_layer_forward since ABI 117), through the C ABI:

* against torch's nn.GRU on the CPU (fp32) and against the library's own per-step path, layer forms x row-tile counts, with and
  without left-over row tiles;
* a whole GRU FullSubNet at a batch whose sub-band rows take the persistent kernels, against the model's tensor-algebra forward
  (itself pinned on the reference's var_gru_b2 golden in test_gpu_family.py).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fsn():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import fullsubnet_amd
    return fullsubnet_amd


def _cus():
    return torch.cuda.get_device_properties(0).multi_processor_count


def _gru_ref(x_tm, params, H):
    """x_tm [T, N, I] (CPU) through torch.nn.GRU on the CPU -> [T, N, H]."""
    I = x_tm.shape[2]
    ref = torch.nn.GRU(I, H, 1)
    with torch.no_grad():
        for name, p in zip(("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"), params):
            getattr(ref, name).copy_(p)
        return ref(x_tm)[0]


def _params(I, H, seed):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / np.sqrt(H)
    mk = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * k * 2.0
    return mk(3 * H, I), mk(3 * H, H), mk(3 * H), mk(3 * H)


# (I, ldx, tiles relative to the CU count, extra tiles): rt = 2 / 3 / 4 plans, one and two K chunks of a narrow input, the stacked
# form (I = H = ldx), with left-over tiles (they advance step by step beside the launch) and without
CASES = [
    (32, 32, 2, 3),     # rt 2 whole round + 3 left-over tiles, two x chunks
    (20, 32, 1, 36),    # fewer workgroups than CUs (tiles / 2), I padded to 32
    (12, 16, 3, 0),     # rt 3, ONE x chunk (Fast FullSubNet's bottleneck width)
    (12, 48, 2, 2),     # rows wider than the padded input: the left-over rows' copy goes step by step
    (32, 32, 4, 4),     # rt 4 + left-over: config 2's plan (1028 tiles on 256 CUs)
    (384, 384, 2, 3),   # stacked layer: input = hidden sequence of the layer below
    (384, 384, 4, 1),
    (384, 384, 3, 0),
]


@pytest.mark.parametrize("I,ldx,per_cu,extra", CASES)
def test_gru_layer_many_rows_on_the_persistent_kernels(fsn, I, ldx, per_cu, extra):
    from fullsubnet_amd import sequence_model as SM
    L = fsn._lib.lib()
    H, T = 384, 7
    tiles = _cus() * per_cu + extra
    N = tiles * 16
    assert L.fsn_gru_layer_is_persistent(T, N, I, ldx, H) == 1
    assert L.fsn_gru_layer_is_persistent(T, 16 * (_cus() + _cus() // 8 - 1), I, ldx, H) == 0  # few rows per CU: step by step
    assert L.fsn_gru_layer_is_persistent(T, N, I, ldx, 320) == 0               # built for 384 units
    torch.manual_seed(I + per_cu)
    params = _params(I, H, seed=3 * I + per_cu)
    x = torch.zeros(T, N, ldx)
    x[:, :, :I] = torch.randn(T, N, I)
    want = _gru_ref(x[:, :, :I].contiguous(), params, H)
    dev = [p.cuda().contiguous() for p in params]
    xd = x.cuda()
    got = SM.gru_layer_infer(xd, *dev)
    torch.cuda.synchronize()
    # the per-step path of the same entry on a slice of the rows (few rows: never persistent)
    sl = slice(N - 16 * 40, N)
    steps = SM.gru_layer_infer(xd[:, sl].contiguous(), *dev).cpu()
    got = got.cpu()
    d_ref = (got - want).abs().max().item()
    d_steps = (got[:, sl] - steps).abs().max().item()
    d_left = (got[:, N - 16 * max(extra, 1):] - want[:, N - 16 * max(extra, 1):]).abs().max().item()
    print(f"GRU layer I = {I} (ldx {ldx}), {tiles} tiles ({per_cu} per CU + {extra}), {T} steps: max |d| vs nn.GRU {d_ref:.2e} "
          f"(last tiles {d_left:.2e}), vs the per-step kernels {d_steps:.2e}; range {want.min():.2f} .. {want.max():.2f}")
    assert torch.isfinite(got).all() and float(want.abs().max()) > 0.3
    assert d_ref <= 2e-5 and d_steps <= 2e-5


def test_gru_layer_many_rows_long_sequence(fsn):
    """190 steps (3 s at config 2's frame rate) with saturating gates: the recurrence's error does not grow with the length."""
    from fullsubnet_amd import sequence_model as SM
    H, T, I = 384, 190, 32
    N = (_cus() * 2 + 1) * 16
    params = _params(I, H, seed=5)
    torch.manual_seed(9)
    x = torch.randn(T, N, I) * 1.5
    want = _gru_ref(x, params, H)
    got = SM.gru_layer_infer(x.cuda(), *[p.cuda().contiguous() for p in params]).cpu()
    d = (got - want).abs()
    print(f"GRU layer, {N} rows, {T} steps: max |d| {d.max():.2e} (first 10 steps {d[:10].max():.2e}, last 10 {d[-10:].max():.2e})")
    assert d.max().item() <= 3e-5


def test_gru_fullsubnet_batch_on_the_persistent_kernels(fsn):
    """A GRU FullSubNet (fullsubnet/model.py:10-70 with sequence_model = "GRU") at a batch whose sub-band rows (B x 257, no band
    dropping) take the persistent kernels: ``_forward_composed_rows`` against the tensor-algebra forward, whose GRU blocks are
    forced onto the per-step path."""
    from fullsubnet_amd import sequence_model as SM
    kw = dict(num_freqs=257, look_ahead=2, sequence_model="GRU", fb_num_neighbors=0, sb_num_neighbors=15,
              fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
              sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
    torch.manual_seed(13)
    m = fsn.Model(**kw)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
    m = m.cuda().eval()
    B, T = 2 * _cus() * 16 // 257 + 2, 21   # a little more than two row tiles per CU
    L = fsn._lib.lib()
    rows_p = (B * 257 + 15) // 16 * 16
    assert L.fsn_gru_layer_is_persistent(T + 2, rows_p, 32, 32, 384) == 1
    mag = (torch.rand(B, 1, 257, T, device="cuda") ** 2) * 3.0
    with torch.no_grad():
        rows = m(mag)
        # reference: utterance by utterance (257 rows each: step by step)
        parts = [m(mag[b:b + 1]) for b in range(0, B, 7)]
    # offline_laplace_norm takes one mean per utterance, so single-utterance calls see the same statistics
    single = torch.cat(parts, 0)
    d = (rows[::7] - single).abs().max().item()
    print(f"GRU FullSubNet, {B} utterances ({rows_p} sub-band rows), {T} frames: max |d| persistent vs per-step {d:.2e}, "
          f"mask range {rows.min():.2f} .. {rows.max():.2f}")
    assert torch.isfinite(rows).all() and float(rows.abs().max()) > 0.05 and d <= 2e-5


def test_gru_fullsubnet_at_config2_size_vs_the_oracle(fsn):
    """BASELINE config 2's shape (64 utterances x 3 s: 16 448 sub-band rows = 4 row tiles on every CU + 4 left-over tiles)
    with sequence_model = "GRU": the compressed mask of two utterances of the batch against the CPU oracle
    (oracle/fullsubnet_oracle.py with cell = "GRU", pinned on the reference's var_gru_b2 golden); bound 1e-4 (north star)."""
    from fsn_synthetic import make_noisy
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.acoustics.feature import stft
    kw = dict(num_freqs=257, look_ahead=2, sequence_model="GRU", fb_num_neighbors=0, sb_num_neighbors=15,
              fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
              sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=True)
    torch.manual_seed(3)
    m = fsn.Model(**kw)
    params = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    B, L = 64, 48000
    noisy_np = np.tile(make_noisy(8, L, seed=1), (B // 8, 1))
    rows = [0, B - 1]
    with torch.no_grad():
        mag = stft(torch.from_numpy(noisy_np).cuda(), 512, 256, 512, return_phase=False)[0]
        assert fsn._lib.lib().fsn_gru_layer_is_persistent(mag.shape[2] + 2, (B * 257 + 15) // 16 * 16, 32, 32, 384) == 1
        first = m(mag.unsqueeze(1))
        again = m(mag.unsqueeze(1))
        # rows never meet inside the kernels and every sum has a fixed order: two runs are bit-identical, left-over tiles on the
        # auxiliary stream or not
        assert torch.equal(first, again)
        got = first[rows].cpu().numpy()
    want = O.fullsubnet_forward(O.stft(noisy_np[rows])[0][:, None], params, cell="GRU", num_groups_in_drop_band=1)
    err = float(np.abs(got - want).max())
    print(f"GRU FullSubNet, 64 x 3 s: max |d| of the compressed mask vs the oracle {err:.2e} (mask range {want.min():.2f} .. {want.max():.2f})")
    assert got.shape == want.shape and np.isfinite(got).all() and float(np.abs(want).max()) > 0.05 and err <= 1e-4


def test_gru_fullsubnet_call_as_a_hip_graph(fsn):
    """fullsubnet_amd.GraphedCall on a GRU FullSubNet whose sub-band rows take the persistent kernels with left-over tiles (the fork
    to the auxiliary stream, the left-over rows' 2-D copies and step launches, the join - all inside the capture): replays are
    bit-identical to the eager call."""
    kw = dict(num_freqs=257, look_ahead=2, sequence_model="GRU", fb_num_neighbors=0, sb_num_neighbors=15,
              fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
              sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=True)
    torch.manual_seed(5)
    m = fsn.Model(**kw).cuda().eval()
    B, T = 2 * _cus() * 16 // 257 + 2, 17
    mag = (torch.rand(B, 1, 257, T, device="cuda") ** 2) * 3.0
    with torch.no_grad():
        eager = m(mag).clone()
    graphed = fsn.GraphedCall(m)
    first = graphed(mag).clone()
    again = graphed(mag).clone()
    assert torch.isfinite(eager).all() and torch.equal(first, eager) and torch.equal(again, eager)
