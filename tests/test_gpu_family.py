"""Parity of the SequenceModel block and the sibling model families (Fast FullSubNet, full-band
baseline) on the HIP kernels against the reference's golden vectors and the CPU oracle.
Needs a real MI355X:  python -m pytest tests -m gpu"""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import model_family_oracle as MF

pytestmark = pytest.mark.gpu

FAST_KW = dict(look_ahead=2, shrink_size=2, sequence_model="LSTM", num_mels=64, encoder_input_size=257,
               bottleneck_hidden_size=384, bottleneck_num_layers=2, noisy_input_num_neighbors=5,
               encoder_output_num_neighbors=0, norm_type="offline_laplace_norm", weight_init=False)


@pytest.fixture(scope="module")
def fsn():
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need a ROCm device")
    import fullsubnet_amd
    fullsubnet_amd._lib.lib()  # raises if libfsn_hip.so is missing: no fallback
    return fullsubnet_amd


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return z, ast.literal_eval(str(z["meta"]))


@pytest.mark.parametrize("name", ["fast_b2_even", "fast_b3_odd"])
def test_fast_fullsubnet_vs_reference(fsn, golden_dir, name):
    from fullsubnet_amd.fast_fullsubnet import Model
    z, meta = load(golden_dir, name)
    params = MF.make_fast_params(seed=meta["seed_w"], gain=meta["gain"])
    m = Model(**FAST_KW)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = torch.from_numpy(z["fb"])
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        crm = m(torch.from_numpy(z["mag"]).cuda().unsqueeze(1)).cpu().numpy()
    assert crm.shape == z["crm"].shape
    assert np.abs(crm - z["crm"]).max() <= 1e-4  # north-star tolerance on the (compressed) mask


def test_fast_fullsubnet_at_baseline_length_vs_reference(fsn, golden_dir):
    """BASELINE config 4's sequence length (3 s, T = 188: 190 steps in the encoder / decoder, 96 in the bottleneck)
    against the reference model's output (tests/golden/fast_long_b2.npz, every 4th bin)."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.fast_fullsubnet import Model
    z, meta = load(golden_dir, "fast_long_b2")
    params = MF.make_fast_params(seed=meta["seed_w"], gain=meta["gain"])
    m = Model(**FAST_KW)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = torch.from_numpy(z["fb"])
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    noisy = torch.from_numpy(O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])).cuda()
    mag = fsn.stft(noisy, 512, 256, 512)[0]
    with torch.no_grad():
        crm = m(mag.unsqueeze(1)).cpu().numpy()[:, :, z["bins"]]
    assert crm.shape == z["crm"].shape
    assert np.abs(crm - z["crm"]).max() <= 1e-4


@pytest.mark.parametrize("batch", [32, 48, 64])
def test_fast_fullsubnet_many_rows_on_the_persistent_kernels(fsn, batch):
    """B = 32 / 48 / 64 -> 2048 / 3072 / 4096 bottleneck rows (not the step kernels of the small golden cases): both
    bottleneck layers as one launch of the group kernel (32 clusters; 64 clusters, two per workgroup set), or - 192
    row tiles - layer by layer on the persistent recurrent kernels; checked against the oracle on 2 utterances."""
    from fullsubnet_amd.fast_fullsubnet import Model
    params = MF.make_fast_params(seed=5)
    m = Model(**FAST_KW)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    rng = np.random.default_rng(0)
    mag = np.abs(rng.standard_normal((batch, 1, 257, 21))).astype(np.float32)
    mag[1] *= 3.0
    with torch.no_grad():
        crm = m(torch.from_numpy(mag).cuda()).cpu().numpy()
        crm2 = m(torch.from_numpy(mag).cuda()).cpu().numpy()
    assert np.array_equal(crm, crm2)
    params["mel_scale.fb"] = m.mel_scale.fb.cpu().numpy()
    want = MF.fast_fullsubnet_forward(mag[[1, batch - 1]], params)
    assert np.abs(crm[[1, batch - 1]] - want).max() <= 1e-4


@pytest.mark.parametrize("batch", [112, 128, 176, 192, 256])
def test_fast_fullsubnet_config4_full_size(fsn, batch):
    """BASELINE config 4's own batch (256 utterances x 64 mel bands = 16 384 bottleneck rows = 1024 row tiles: FOUR tiles
    per workgroup on the persistent recurrent kernels - layer 0 `lstm_rec_kernel<384,4,2,true>` on the generic x_rows
    input, layer 1 `lstm_rec_x_kernel<384,4,2,0,true>` with the hidden sequence streamed out, 81 % of that config's
    step), the two neighbouring plans (128 -> RT = 2, 192 -> RT = 3) and two batches between the plans, whose rows the model
    pads to a count the persistent kernels take whole (fsn_lstm_layer_plan_rows: 112 utterances = 448 tiles as 224
    workgroups x 2, 176 = 704 -> 705 tiles as 235 x 3): the oracle on two utterances (1e-4, the
    north-star bound on the compressed mask), and EVERY utterance against its own result inside a batch of 32 (group
    kernel: other instantiations, other summation order).  fast_fullsubnet/model.py:143-202."""
    from fullsubnet_amd.fast_fullsubnet import Model
    params = MF.make_fast_params(seed=7)
    m = Model(**FAST_KW)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    rng = np.random.default_rng(batch)
    mag = np.abs(rng.standard_normal((batch, 1, 257, 43))).astype(np.float32)
    mag *= rng.uniform(0.3, 3.0, (batch, 1, 1, 1)).astype(np.float32)
    x = torch.from_numpy(mag).cuda()
    with torch.no_grad():
        crm = m(x).cpu().numpy()
        crm2 = m(x).cpu().numpy()
        small = torch.cat([m(x[i:i + 32]) for i in range(0, batch, 32)], dim=0).cpu().numpy()
        # the bottleneck's last layer with the kernel's fused output layer (fsn_lstm_layer_forward_fc: no hidden sequence
        # written) against the layer + fsn_linear_forward form
        units_rows = fsn._lib.lib().fsn_lstm_layer_plan_rows(batch * 64, 384)
        assert units_rows >= batch * 64 and (units_rows > batch * 64) == (batch == 176)
        assert fsn._lib.lib().fsn_lstm_layer_fc_supported(22, units_rows, 384, 384, 384, 1) == 1
        m.fused_output_layer = False
        unfused = m(x).cpu().numpy()
        m.fused_output_layer = True
    assert np.abs(crm - unfused).max() <= 2e-5
    assert np.isfinite(crm).all() and np.array_equal(crm, crm2)
    d_plan = np.abs(crm - small).reshape(batch, -1).max(axis=1)
    params["mel_scale.fb"] = m.mel_scale.fb.cpu().numpy()
    pick = [0, batch // 2 + 1, batch - 1]
    want = MF.fast_fullsubnet_forward(mag[pick], params)
    d_oracle = np.abs(crm[pick] - want).max()
    print(f"fast B={batch}: max|d cIRM| vs oracle {d_oracle:.2e}, vs the batch-32 plan {d_plan.max():.2e}, "
          f"mask range {crm.min():.2f} .. {crm.max():.2f}")
    assert np.abs(crm).max() > 1.0  # the 1e-4 bound means something
    assert d_oracle <= 1e-4
    assert d_plan.max() <= 5e-5


@pytest.mark.parametrize("batch,frames,shrink,nn_noisy,nn_enc", [(3, 21, 2, 5, 0), (5, 22, 2, 5, 0), (2, 20, 3, 2, 1),
                                                                 (17, 9, 4, 0, 3), (1, 3, 2, 5, 0)])
def test_fast_fullsubnet_glue_kernels_vs_the_tensor_algebra(fsn, batch, frames, shrink, nn_noisy, nn_enc):
    """The inference forward with time-major tensors and the glue on fast_glue_kernels.hip (look-ahead pad, norms, unit
    windows with reflected edges, real_time_down / up-sampling incl. a shorter last block, cats, the final reshape) against
    the composed forward (the reference's own tensor algebra, model.py:108-140, 143-202) on the same LSTM kernels, and
    against the numpy oracle: other shrink sizes and neighbour counts than the shipped TOML's, rows that need padding."""
    from fullsubnet_amd.fast_fullsubnet import Model
    kw = dict(FAST_KW, shrink_size=shrink, noisy_input_num_neighbors=nn_noisy, encoder_output_num_neighbors=nn_enc)
    params = MF.make_fast_params(seed=11, nn_noisy=nn_noisy, nn_enc=nn_enc)
    m = Model(**kw)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    rng = np.random.default_rng(frames)
    mag = np.abs(rng.standard_normal((batch, 1, 257, frames))).astype(np.float32)
    mag *= rng.uniform(0.3, 3.0, (batch, 1, 1, 1)).astype(np.float32)
    x = torch.from_numpy(mag).cuda()
    with torch.no_grad():
        assert m._rows_path_ok(x)
        got = m(x)
        m.rows_path = False
        composed = m(x)
        m.rows_path = True
    assert got.shape == composed.shape == (batch, 2, 257, frames)
    scale = float(composed.abs().max())
    d = float((got - composed).abs().max())
    print(f"fast glue B={batch} T={frames} s={shrink}: max|d| vs the composed forward {d:.2e} (mask scale {scale:.2f})")
    assert d <= 2e-5 * max(1.0, scale)
    params["mel_scale.fb"] = m.mel_scale.fb.cpu().numpy()
    want = MF.fast_fullsubnet_forward(mag[[0, batch - 1]], params, look_ahead=kw["look_ahead"], shrink_size=shrink,
                                      noisy_input_num_neighbors=nn_noisy, encoder_output_num_neighbors=nn_enc)
    assert np.abs(got.cpu().numpy()[[0, batch - 1]] - want).max() <= 1e-4


def test_fullband_baseline_vs_reference(fsn, golden_dir):
    from fullsubnet_amd.fullband_baseline import Model
    z, meta = load(golden_dir, "fullband_b2")
    params = MF.make_fullband_params(seed=meta["seed_w"], gain=meta["gain"])
    m = Model(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=None, look_ahead=2,
              norm_type="offline_laplace_norm", weight_init=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        crm = m(torch.from_numpy(z["mag"]).cuda().unsqueeze(1)).cpu().numpy()
    assert np.abs(crm - z["crm"]).max() <= 1e-4


@pytest.mark.parametrize("cell", ["LSTM", "GRU"])
@pytest.mark.parametrize("I,H,O,layers,act,B,T", [
    (64, 384, 0, 1, None, 3, 9),       # no output layer (Fast FullSubNet encoder.0)
    (40, 257, 64, 1, "ReLU", 5, 7),    # hidden size padded 257 -> 320 (encoder.1)
    (12, 384, 1, 2, "ReLU", 130, 6),   # bottleneck shape, rows not a multiple of 16
    (20, 64, 8, 2, "Tanh", 2, 5),
    (33, 128, 5, 1, None, 1100, 4),    # >= 64 row tiles: multi-tile step / BPTT kernels
])
def test_sequence_model_inference_and_training_vs_torch(fsn, cell, I, H, O, layers, act, B, T):
    """SequenceModel.forward under no_grad (inference kernels) and under autograd (forward with saves +
    BPTT) against ATen's nn.LSTM / nn.Linear on CPU with the same parameters."""
    from fullsubnet_amd.sequence_model import SequenceModel
    torch.manual_seed(I + H)
    m = SequenceModel(I, O, H, layers, False, cell, act)
    ref_lstm = (torch.nn.LSTM if cell == "LSTM" else torch.nn.GRU)(I, H, layers, batch_first=True)
    ref_lstm.load_state_dict(m.sequence_model.state_dict())
    x = torch.randn(B, I, T)
    dy = torch.randn(B, O or H, T)

    def ref_forward(xin):
        o, _ = ref_lstm(xin.permute(0, 2, 1))
        if O:
            o = torch.nn.functional.linear(o, m.fc_output_layer.weight.detach().cpu(), m.fc_output_layer.bias.detach().cpu())
        if act:
            o = {"ReLU": torch.relu, "Tanh": torch.tanh}[act](o)
        return o.permute(0, 2, 1)

    xr = x.clone().requires_grad_(True)
    yr = ref_forward(xr)
    (yr * dy).sum().backward()
    md = m.cuda()
    with torch.no_grad():
        yi = md(x.cuda())
    assert (yi.cpu() - yr.detach()).abs().max().item() <= 2e-5
    xd = x.cuda().requires_grad_(True)
    yt = md(xd)
    (yt * dy.cuda()).sum().backward()
    assert (yt.detach().cpu() - yr.detach()).abs().max().item() <= 2e-5
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= 1e-4 * max(xr.grad.abs().max().item(), 1.0)
    for k in range(layers):
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            g = getattr(md.sequence_model, f"{n}_l{k}").grad.cpu()
            r = getattr(ref_lstm, f"{n}_l{k}").grad
            assert (g - r).abs().max().item() <= 1e-4 * max(r.abs().max().item(), 1.0), (n, k)



@pytest.mark.parametrize("I,O,B", [(12, 1, 8192), (24, 0, 8192), (28, 1, 12288), (12, 0, 8300), (24, 1, 8300), (28, 0, 12400),
                                   (32, 1, 16448), (24, 1, 24688)])
def test_stacked_lstm_on_the_persistent_kernels_vs_torch(fsn, I, O, B):
    """A two-layer H = 384 stack with a narrow input on enough rows for the persistent kernels (512 / 768 row tiles: two /
    three per workgroup, nothing left over): layer 0 on lstm_rec_in_kernel's row-major form with one (I <= 16) or two K
    chunks of input, layer 1 on lstm_rec_x_kernel (hidden sequence out, or the fused output layer); 8300 / 12400 / 16448 rows
    (whole rounds of two / three / four tiles per workgroup + 7 / 7 / 4 left-over tiles): the same two kernels on the whole
    rounds, the left-over rows step by step beside them from their own small projection (round 6; before: the full projection
    GEMM + lstm_rec_kernel); 24688 rows = 1543 tiles: two rounds of three tiles per workgroup + 7 left over (before: one round of
    five and 263 tiles step by step).  Against ATen's nn.LSTM / nn.Linear on the CPU (sequence_model.py:52-58)."""
    from fullsubnet_amd.sequence_model import SequenceModel
    torch.manual_seed(I + B)
    T, H = 5, 384
    m = SequenceModel(I, O, H, 2, False, "LSTM", "ReLU" if O else None)
    ref = torch.nn.LSTM(I, H, 2, batch_first=True)
    ref.load_state_dict(m.sequence_model.state_dict())
    x = torch.randn(B, I, T)
    with torch.no_grad():
        want, _ = ref(x.permute(0, 2, 1))
        if O:
            want = torch.relu(torch.nn.functional.linear(want, m.fc_output_layer.weight, m.fc_output_layer.bias))
        got = m.cuda()(x.cuda()).cpu()
    assert got.shape == (B, O or H, T)
    assert (got - want.permute(0, 2, 1)).abs().max().item() <= 2e-5

@pytest.mark.parametrize("rows", [(640, 800, 192, 128), (630, 790, 170, 100), (1500, 24)])
def test_several_sequence_models_in_one_persistent_launch_vs_torch(fsn, rows):
    """sequence_model.multi_forward (fsn_lstm2_forward_multi: one launch of the group kernel, a weight set per model -
    the band sections of improved_fullsubnet/model.py:402-449 at batch 32) against ATen's nn.LSTM / nn.Linear on the CPU
    and against each model on its own; row counts that are not multiples of 64 / 16, different input widths."""
    from fullsubnet_amd.sequence_model import SequenceModel, multi_forward, multi_plan
    torch.manual_seed(sum(rows))
    T, H = 9, 384
    widths = (62, 68, 100, 180)[:len(rows)]
    outs_n = (2, 8, 40, 120)[:len(rows)]
    models = [SequenceModel(i, o, H, 2, False, "LSTM", None) for i, o in zip(widths, outs_n)]
    xs = [torch.randn(n, i, T) for n, i in zip(rows, widths)]
    refs = []
    for m, x in zip(models, xs):
        ref = torch.nn.LSTM(m.input_size, H, 2, batch_first=True)
        ref.load_state_dict(m.sequence_model.state_dict())
        with torch.no_grad():
            o, _ = ref(x.permute(0, 2, 1))
            refs.append(torch.nn.functional.linear(o, m.fc_output_layer.weight, m.fc_output_layer.bias).permute(0, 2, 1))
    models = [m.cuda() for m in models]
    xd = [x.cuda() for x in xs]
    assert multi_plan(models, [tuple(x.shape) for x in xs]), "these shapes are meant to take the persistent launch"
    with torch.no_grad():
        outs = multi_forward(models, xd)
        single = [m(x) for m, x in zip(models, xd)]
    from fullsubnet_amd import _lib
    assert _lib.stream_status(synchronize=True) == (0, 0)
    for o, r, s1 in zip(outs, refs, single):
        assert o.shape == r.shape and torch.isfinite(o).all()
        assert (o.cpu() - r).abs().max().item() <= 2e-5
        assert (o - s1).abs().max().item() <= 2e-5
    # too few rows for the persistent launch: the same entry runs stack by stack
    few = [x[:40].contiguous() for x in xd]
    assert not multi_plan(models, [tuple(x.shape) for x in few])
    with torch.no_grad():
        outs = multi_forward(models, few)
    for o, r in zip(outs, refs):
        assert (o.cpu() - r[:40]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("n_fft,hop", [(512, 128), (960, 480), (400, 100), (1536, 384)])
def test_stft_istft_other_transform_shapes(fsn, golden_dir, n_fft, hop):
    """fsn_stft / fsn_istft on the direct-DFT path (every shape but 512 / 256) against the oracle
    (correctly rounded transform) and torch.stft / torch.istft as the reference calls them."""
    from oracle import fullsubnet_oracle as O
    z, meta = load(golden_dir, "stft_generic")
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    k = f"{n_fft}_{hop}"
    win = z["win/" + k]
    mag, _, re, im = fsn.stft(torch.from_numpy(noisy).cuda(), n_fft, hop, n_fft)
    re, im, mag = re.cpu().numpy(), im.cpu().numpy(), mag.cpu().numpy()
    omag, _, ore, oim = O.stft(noisy, n_fft, hop, n_fft, window=win)
    fmax = np.maximum(np.abs(ore), np.abs(oim)).max(axis=1, keepdims=True)
    ulp = np.spacing(fmax.astype(np.float32))
    assert (np.abs(re - ore) / ulp).max() <= 1.0 and (np.abs(im - oim) / ulp).max() <= 1.0
    scale = np.abs(z["mag/" + k]).max()
    assert np.abs(re - z["re/" + k]).max() <= 2e-6 * scale and np.abs(im - z["im/" + k]).max() <= 2e-6 * scale
    assert np.abs(mag - z["mag/" + k]).max() <= 2e-6 * scale
    r, i = z["re/" + k], z["im/" + k]
    fr = torch.from_numpy(r * np.float32(0.5)).cuda()
    fi = torch.from_numpy(i * np.float32(0.5) + r * np.float32(0.25)).cuda()
    back = fsn.istft((fr, fi), n_fft, hop, n_fft, length=meta["length"], input_type="real_imag").cpu().numpy()
    assert np.abs(back - z["back/" + k]).max() <= 3e-6 * np.abs(z["back/" + k]).max()
    # round trip
    y = torch.from_numpy(noisy).cuda()
    _, _, re2, im2 = fsn.stft(y, n_fft, hop, n_fft)
    rt = fsn.istft((re2, im2), n_fft, hop, n_fft, length=meta["length"], input_type="real_imag")
    assert (rt - y).abs().max().item() <= 2e-6


@pytest.mark.parametrize("name,cfg", [("improved_16k_b2", MF.IMPROVED_16K), ("improved_48k_b2", MF.IMPROVED_48K),
                                      ("improved_769_b2", MF.IMPROVED_48K_769)])
def test_improved_fullsubnet_vs_reference(fsn, golden_dir, name, cfg):
    """Waveform in -> waveform out (stft 512/128 or 960/480 -> fb LSTM -> banded sb LSTMs -> mask -> istft)."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.improved_fullsubnet import Model
    z, meta = load(golden_dir, name)
    params = MF.make_improved_params(cfg, seed=meta["seed_w"])
    m = Model(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    with torch.no_grad():
        enh = m(torch.from_numpy(noisy).cuda().unsqueeze(1)).cpu().numpy()
    assert enh.shape == z["enhanced"].shape
    assert np.abs(enh - z["enhanced"]).max() <= 1e-4 * np.abs(z["enhanced"]).max()


@pytest.mark.parametrize("cfg,batch,length", [(MF.IMPROVED_16K, 3, 5000), (MF.IMPROVED_48K, 2, 24000), (MF.IMPROVED_48K, 32, 9600),
                                              (MF.IMPROVED_48K_769, 1, 15000)])
def test_improved_fullsubnet_glue_kernels_vs_the_tensor_algebra(fsn, cfg, batch, length):
    """Model._forward_kernels (round 5: |X| ** fdrc + last-bin slice, the transposes around the full-band model, the sections'
    outputs re-ordered / padded / multiplied into the two planes - fsn_improved_front, fsn_bft_to_rows, fsn_rows_to_bft,
    fsn_improved_mask_apply) against the same forward as tensor algebra of the host framework (`glue_kernels = False`): pure
    data movement and the same IEEE operations, so BIT-identical - on chain-kernel plans (one to three utterances) and on the
    persistent multi-section launch (batch 32); SequenceModel.forward's own transposes against a permute; both entries'
    argument checks."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.improved_fullsubnet import Model
    from fullsubnet_amd.sequence_model import from_rows, to_rows
    m = Model(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in MF.make_improved_params(cfg, seed=5).items()}, strict=True)
    m = m.cuda().eval()
    y = torch.from_numpy(O.make_noisy(batch, length, seed=9)).cuda()
    with torch.no_grad():
        assert m._glue_on_kernels(y, None)
        got = m(y)
        m.glue_kernels = False
        assert not m._glue_on_kernels(y, None)
        want = m(y)
        m.glue_kernels = True
    assert got.shape == want.shape == (batch, 1, length) and torch.equal(got, want)
    assert float(want.abs().max()) > 0
    x = torch.randn(5, 37, 29, device="cuda")
    h = to_rows(x)
    assert h.shape == (29, 16, 48) and torch.equal(h[:, :5, :37], x.permute(2, 0, 1)) and float(h[:, 5:].abs().max()) == 0
    assert float(h[:, :, 37:].abs().max()) == 0
    assert torch.equal(from_rows(h, 5)[:, :37], x) and torch.equal(from_rows(h[:, :, :20], 5), x[:, :20])
    L = fsn._lib.lib()
    assert L.fsn_bft_to_rows(fsn._lib.dev_ptr(x), 5, 37, 29, fsn._lib.dev_ptr(h), 4, 48, None) != 0  # fewer padded rows than rows
    assert L.fsn_improved_mask_apply(0, None, None, None, 1, 1, 1, None, None, None) != 0


@pytest.mark.parametrize("batch", [1, 3])
def test_improved_fullsubnet_wide_neighbourhood_on_the_kernel_glue(fsn, batch):
    """A section whose unfolded width (8 + 2 x 60 twice = 256 columns) is beyond fsn_improved_section_input's 240-column tile:
    `_section_prepared` declines, the section's input goes through the unfolded tensor - and, on the kernel-glue forward, from
    there into the LSTM entries' own time-major layout (that branch used to hand the wrapped [B, 2, n c, T] tensor
    to fsn_improved_mask_apply, which refused it).  Bit-identical to the tensor-algebra forward."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.improved_fullsubnet import Model
    cfg = dict(MF.IMPROVED_16K, sb_num_neighbor_freqs=[15, 15, 60], fb_num_neighbor_freqs=[15, 15, 60])
    m = Model(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in MF.make_improved_params(cfg, seed=6).items()}, strict=True)
    m = m.cuda().eval()
    y = torch.from_numpy(O.make_noisy(batch, 4000, seed=10)).cuda()
    with torch.no_grad():
        assert m._glue_on_kernels(y, None)
        got = m(y)
        m.glue_kernels = False
        want = m(y)
    assert got.shape == want.shape == (batch, 1, 4000) and torch.equal(got, want) and float(want.abs().max()) > 0


def test_row_transposes_beyond_one_grid_of_rows(fsn):
    """fsn_bft_to_rows / fsn_rows_to_bft with more than 65535 rows (the row index is grid.z: ADVICE r5 - B F >= 65536 rows of a
    composed model used to raise): several launches, same result as a permute."""
    from fullsubnet_amd.sequence_model import from_rows, to_rows
    x = torch.randn(70000, 3, 5, device="cuda")
    h = to_rows(x)
    assert h.shape == (5, 70000, 16) and torch.equal(h[:, :, :3], x.permute(2, 0, 1)) and float(h[:, :, 3:].abs().max()) == 0
    assert torch.equal(from_rows(h, 70000)[:, :3], x)
    assert torch.equal(to_rows(x.half()[:100]), to_rows(x.half()[:100].float()))  # other dtypes are cast, as the strided copy did


def test_improved_fullsubnet_at_baseline_length_vs_reference(fsn, golden_dir):
    """BASELINE config 5's clip: 3 s at 48 kHz through the reference's own 481-bin example (301 frames) against the
    reference model's output (tests/golden/improved_48k_long_b1.npz, every 8th sample)."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.improved_fullsubnet import Model
    z, meta = load(golden_dir, "improved_48k_long_b1")
    cfg = MF.IMPROVED_48K
    params = MF.make_improved_params(cfg, seed=meta["seed_w"])
    m = Model(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    with torch.no_grad():
        enh = m(torch.from_numpy(noisy).cuda().unsqueeze(1)).cpu().numpy()[..., ::meta["sample_stride"]]
    assert enh.shape == z["enhanced"].shape
    assert np.abs(enh - z["enhanced"]).max() <= 1e-4 * float(z["enhanced_absmax"])


@pytest.mark.parametrize("cfg", [MF.IMPROVED_16K, MF.IMPROVED_48K, MF.IMPROVED_48K_769])
def test_improved_section_input_kernel_vs_the_unfolded_tensor(fsn, cfg):
    """fsn_improved_section_input (gather + offline Laplace norm + time-major layout, the unfolded tensor never formed)
    against the reference's operation sequence as tensor algebra (model.py:402-440: two `_freq_unfold`s, cat, norm) for
    every section, whole and on a unit range (the norm statistic is the whole section's either way)."""
    from fullsubnet_amd.improved_fullsubnet import Model
    torch.manual_seed(3)
    m = Model(**cfg).cuda().eval()
    sb = m.sb_model
    B, F, T = 3, cfg["num_freqs"] - 1, 77
    noisy = torch.rand(B, 1, F, T, device="cuda") + 0.05
    fb = torch.randn(B, 1, F, T, device="cuda")
    worst = 0.0
    with torch.no_grad():
        for i, n_units in enumerate(sb.num_units(F)):
            for units in (None, (1, n_units), (0, 1)):
                if units is not None and units[1] > n_units:
                    continue
                want = sb._section_input(noisy, fb, i, units)  # [B, n, 1, W, T]
                h, rows = sb._section_prepared(noisy, fb, i, units)
                n, W = want.shape[1], want.shape[3]
                assert rows == B * n and h.shape[0] == T and h.shape[1] % 16 == 0 and h.shape[2] % 16 == 0
                got = h[:, :rows, :W].permute(1, 2, 0).reshape(B, n, 1, W, T)
                worst = max(worst, ((got - want).abs().max() / want.abs().max()).item())
                assert torch.equal(h[:, rows:], torch.zeros_like(h[:, rows:]))
                assert torch.equal(h[:, :, W:], torch.zeros_like(h[:, :, W:]))
    print(f"section input kernel vs tensor algebra: max relative difference {worst:.3g}")
    assert worst <= 5e-7


def test_improved_fullsubnet_config5_full_size_on_the_persistent_launch(fsn, golden_dir):
    """BASELINE config 5 at its full size - 32 utterances x 3 s at 48 kHz - where the four band sections run as ONE
    persistent launch of the group kernel with a weight set per section (fsn_lstm2_forward_multi).  The reference's
    utterance of improved_48k_long_b1 sits at both ends of the batch: both copies must match the reference model's
    output; and, a size-independent property (the model has no cross-utterance term), every utterance must equal its
    result in a batch of 16, which runs on other kernels (wavefronts on one stream per section)."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd import _lib
    from fullsubnet_amd.improved_fullsubnet import Model
    z, meta = load(golden_dir, "improved_48k_long_b1")
    cfg = MF.IMPROVED_48K
    params = MF.make_improved_params(cfg, seed=meta["seed_w"])
    m = Model(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    assert meta["batch"] == 1
    gold = O.make_noisy(1, meta["length"], seed=meta["seed_x"])
    rest = O.make_noisy(30, meta["length"], seed=1234)
    noisy = torch.from_numpy(np.concatenate([gold, rest, gold], axis=0)).cuda()
    before = _lib.persist_stats()[0]
    with torch.no_grad():
        full = m(noisy.unsqueeze(1))
    assert _lib.stream_status(synchronize=True) == (0, 0)
    assert _lib.persist_stats()[0] - before == 2, "full-band chain + ONE launch for the four sections"
    assert torch.isfinite(full).all()
    stride, ref, scale = meta["sample_stride"], z["enhanced"], float(z["enhanced_absmax"])
    for b in (0, 31):
        got = full[b:b + 1].cpu().numpy()[..., ::stride]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-4 * scale, b
    with torch.no_grad():
        halves = torch.cat([m(noisy[:16].unsqueeze(1)), m(noisy[16:].unsqueeze(1))], dim=0)
    err = (full - halves).abs().max().item()
    print(f"config 5, 32 utterances on the persistent launch vs 2 x 16 on wavefronts: max |d| {err:.3g} "
          f"(output scale {full.abs().max().item():.3g})")
    assert err <= 5e-6 * full.abs().max().item()  # measured 8.6e-7 of the output scale


def test_improved_fullsubnet_batches_beyond_one_persistent_launch_run_as_chunks(fsn):
    """More utterances than one persistent launch of the band sections holds (32 clusters of 64 rows: 33 utterances at 48
    kHz): Model.forward runs the batch as chunks that fit plus a remainder.  Every utterance must equal its result in a
    batch of 12 (wavefronts on one stream per section)."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.improved_fullsubnet import Model
    cfg = MF.IMPROVED_48K
    m = Model(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in MF.make_improved_params(cfg, seed=2).items()}, strict=True)
    m = m.cuda().eval()
    noisy = torch.from_numpy(O.make_noisy(40, 14400, seed=77)).cuda()
    T = 1 + 14400 // cfg["hop_length"]
    chunk = m._persistent_chunk(40, T)
    assert chunk is not None and 28 <= chunk < 40, chunk
    with torch.no_grad():
        whole = m(noisy)
        parts = torch.cat([m(noisy[i:i + 12]) for i in range(0, 40, 12)], dim=0)
    assert whole.shape == (40, 1, 14400) and torch.isfinite(whole).all()
    assert (whole - parts).abs().max().item() <= 5e-6 * parts.abs().max().item()


@pytest.mark.parametrize("world", [3, 8])
@pytest.mark.parametrize("name,cfg", [("improved_16k_b2", MF.IMPROVED_16K), ("improved_48k_b2", MF.IMPROVED_48K),
                                      ("improved_769_b2", MF.IMPROVED_48K_769)])
def test_improved_fullsubnet_unit_shard_vs_reference(fsn, golden_dir, name, cfg, world):
    """The frequency-axis shard of BASELINE config 5 (SubbandModel.forward_units: every rank runs its share of each
    section's units, at 8 ranks some sections leave ranks without any) with the ranks played one after the other on
    this GPU and the all-gather replaced by a concatenation in rank order - what parallel.gather_ragged produces
    (tests/test_parallel_cpu.py holds the collective itself to that).  Same tolerance as the unsharded model."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.improved_fullsubnet import Model
    z, meta = load(golden_dir, name)
    params = MF.make_improved_params(cfg, seed=meta["seed_w"])
    m = Model(**cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    sb = m.sb_model
    seen = []

    def ranks_in_turn(noisy_mag, fb_output, unit_group=None):
        per_rank = [sb.forward_units(noisy_mag, fb_output, r, world) for r in range(world)]
        seen.append([[int(t.shape[0]) for t in parts] for parts in per_rank])
        return sb.assemble_units([torch.cat(sec, dim=0) for sec in zip(*per_rank)])

    sb.forward = ranks_in_turn
    m.glue_kernels = False  # (a real unit shard passes `unit_group`, which takes the tensor-algebra forward that calls sb.forward)
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    with torch.no_grad():
        enh = m(torch.from_numpy(noisy).cuda().unsqueeze(1)).cpu().numpy()
    units = sb.num_units(cfg["num_freqs"] - 1)
    assert [sum(col) for col in zip(*seen[0])] == units
    assert enh.shape == z["enhanced"].shape
    assert np.abs(enh - z["enhanced"]).max() <= 1e-4 * np.abs(z["enhanced"]).max()


@pytest.mark.parametrize("name", ["var_gru_b2", "var_gaussian_b2", "var_cln_b2", "var_forgetting_b2",
                                  "var_fbnn2_tanh_b3", "var_gru_b33"])
def test_fullsubnet_constructor_variants_vs_reference(fsn, golden_dir, name):
    """FullSubNet with the constructor options outside the shipped TOMLs (GRU, extra norms, fb neighbours,
    other activations): composed from SequenceModel blocks on the HIP kernels, against the reference model."""
    from oracle import fullsubnet_oracle as O
    z, meta = load(golden_dir, name)
    kw = meta["kw"]
    params = O.make_params(seed=meta["seed_w"], gain=meta["gain"], mask_gain=meta["mask_gain"], gates=meta["gates"],
                           fb_num_neighbors=kw["fb_num_neighbors"])
    m = fsn.Model(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        crm = m(torch.from_numpy(z["mag"]).cuda().unsqueeze(1)).cpu().numpy()
    assert crm.shape == z["crm"].shape
    if name == "var_gru_b33":  # the reference's own GRU model at a batch whose sub-band rows take the persistent many-row kernels
        L = fsn._lib.lib()
        assert L.fsn_gru_layer_is_persistent(z["mag"].shape[2] + 2, (33 * 257 + 15) // 16 * 16, 32, 32, 384) == 1
        print(f"var_gru_b33: max |d| {np.abs(crm - z['crm']).max():.2e} of a mask range {np.abs(z['crm']).max():.1f}")
    assert np.abs(crm - z["crm"]).max() <= 1e-4  # measured 1.0e-5 .. 2.9e-5


@pytest.mark.parametrize("kw,B", [
    (dict(sequence_model="GRU"), 3),
    (dict(sequence_model="GRU", norm_type="cumulative_laplace_norm"), 1),
    (dict(fb_model_hidden_size=256, sb_model_hidden_size=192, fb_output_activate_function="Tanh"), 3),
    (dict(sb_num_neighbors=7, sb_model_hidden_size=320, norm_type="cumulative_laplace_norm", num_groups_in_drop_band=3), 4),
    (dict(sb_output_activate_function="Tanh", num_groups_in_drop_band=1), 2),
])
def test_composed_variants_with_the_glue_on_the_library(fsn, kw, B):
    """Constructor variants outside the fused kernels' specialisation (GRU, other hidden sizes, other output activations;
    fullsubnet/model.py:10-70) in inference: ``_forward_composed_rows`` keeps every tensor between the two SequenceModel blocks
    time-major and forms it with the library's glue kernels (pad + norm, unfold ++ full-band output + norm + drop_band, reshape
    + look-ahead slice) - held to ``_forward_composed``, the reference's forward operation by operation as tensor algebra
    (itself held to the reference's goldens: test_fullsubnet_constructor_variants_vs_reference), incl. the eval-mode band
    dropping of batches (quirk Q1), both shipped norms, one utterance."""
    base = dict(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=2, weight_init=False)
    torch.manual_seed(11)
    m = fsn.Model(**{**base, **kw})
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)  # masks of a few units, like the goldens' weights
    m = m.cuda().eval()
    assert not m._fused
    mag = (torch.rand(B, 1, 257, 37, device="cuda") ** 2) * 3.0
    with torch.no_grad():
        assert m._composed_rows_ok(mag)
        rows = m(mag)
        m.composed_rows = False
        algebra = m(mag)
    assert rows.shape == algebra.shape and torch.isfinite(rows).all()
    assert float(algebra.abs().max()) > 0.05
    assert (rows - algebra).abs().max().item() <= 2e-5  # measured 1e-6 .. 4e-6


@pytest.mark.parametrize("name", ["fast_train_b3", "fullband_train_b3"])
def test_sibling_training_step_vs_the_reference(fsn, golden_dir, name):
    """ONE training step of the two sibling recipes that ship training TOMLs (fast_fullsubnet/train_shrinkSize2.toml:69-79 +
    fast_fullsubnet/trainer.py:33-76; fullband_baseline/train.toml:70-78 + its trainer.py:32-71) against the REFERENCE's own
    step (tests/golden/make_golden_family_train.py, fp32): loss, what clip_grad_norm_ returns, every parameter tensor's clipped
    gradient (norm and strided samples) and the Adam-updated parameters.  Here the LSTM / Linear blocks run on the library's
    training entries and the glue between them is autograd-tracked tensor algebra.  Fast FullSubNet's mel
    filterbank is the restated torchaudio one on both sides (parity unpinned at that boundary only)."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.train import train_step
    z, meta = load(golden_dir, name)
    if meta["model"] == "fast_fullsubnet":
        from fullsubnet_amd.fast_fullsubnet import Model
        params = MF.make_fast_params(seed=meta["seed_w"], gain=meta["gain"])
        m = Model(**FAST_KW)
        sd = {k: torch.from_numpy(v) for k, v in params.items()}
        sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    else:
        from fullsubnet_amd.fullband_baseline import Model
        params = MF.make_fullband_params(seed=meta["seed_w"], gain=meta["gain"], out_gain=meta["out_gain"])
        m = Model(num_freqs=257, hidden_size=512, sequence_model="LSTM", output_activate_function=False, look_ahead=2,
                  norm_type="offline_laplace_norm", weight_init=False)
        sd = {k: torch.from_numpy(v) for k, v in params.items()}
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    noisy = torch.from_numpy(O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_noisy"])).cuda()
    clean = torch.from_numpy((meta["clean_gain"] * O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_clean"]))
                             .astype(np.float32)).cuda()
    opt = fsn.ClipAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    loss = train_step(m, opt, noisy, clean).item()
    rel_loss = abs(loss - float(z["loss"])) / float(z["loss"])
    rel_total = abs(float(opt.total_norm) - float(z["total_norm"])) / float(z["total_norm"])
    s = meta["sample"]
    worst_norm, worst_elem, moved = ("", 0.0), ("", 0.0), 0
    for k, p in m.named_parameters():
        gn = float(z["gnorm/" + k])
        worst_norm = max(worst_norm, (k, abs(float(p.grad.norm()) - gn) / (gn + 1e-30)), key=lambda kv: kv[1])
        g = p.grad.detach().reshape(-1)[::s].cpu().numpy()
        r = z["g/" + k]
        worst_elem = max(worst_elem, (k, float(np.abs(g - r).max() / max(np.abs(r).max(), 1e-3 * gn, 1e-30))), key=lambda kv: kv[1])
        pv = p.detach().reshape(-1)[::s].cpu().numpy()
        firm = np.abs(r) > 1e-5
        if firm.any():
            assert np.abs(pv - z["p/" + k])[firm].max() <= 2.1e-3, k  # a sign flip of a firm gradient would be 2 lr
            moved += int((np.abs(pv - params[k].reshape(-1)[::s]) > 5e-4).sum())
    print(f"{name}: loss {rel_loss:.2e}, total norm {rel_total:.2e}, worst tensor norm {worst_norm[1]:.2e} ({worst_norm[0]}), "
          f"worst sampled element {worst_elem[1]:.2e} of the tensor's max ({worst_elem[0]})")
    # measured r06 (printed); bounds ~3x above
    assert rel_loss <= 2e-6 and rel_total <= 5e-5 and worst_norm[1] <= 5e-4 and worst_elem[1] <= 2e-3
    assert moved > 0


def _fast_train_model(fsn, meta, arith):
    from fullsubnet_amd.fast_fullsubnet import Model
    params = MF.make_fast_params(seed=meta["seed_w"], gain=meta["gain"])
    m = Model(**FAST_KW)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    sd["mel_scale.fb"] = m.mel_scale.fb.clone()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    m.train_arithmetic = arith
    return m


def _step_distances(model, opt, loss, z, sample):
    """(loss, total gradient norm, worst parameter tensor's gradient norm) of a step, relative to a golden step's."""
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        gn = float(z["gnorm/" + k])
        worst = max(worst, (k, abs(float(p.grad.norm()) - gn) / (gn + 1e-30)), key=lambda kv: kv[1])
    return (abs(loss - float(z["loss"])) / float(z["loss"]),
            abs(float(opt.total_norm) - float(z["total_norm"])) / float(z["total_norm"]), worst)


def test_fast_fullsubnet_amp_step_vs_the_references_own_fp16_autocast_step(fsn, golden_dir):
    """fast_fullsubnet/train_shrinkSize2.toml:5 trains with use_amp = true.  Under the trainer's 16-bit arithmetic
    (Model.train_arithmetic = "f16") the bottleneck - two LSTM layers x 384 units on B x 64 rows, 90 % of the step's products -
    runs on the persistent 16-bit training kernels in pieces of whole clusters (fullsubnet_amd.train.lstm2_train_chunks; here 24
    utterances = ONE piece of 1536 rows; the shipped batch of 72 = three).  Arbiters, both the reference's own steps at this
    shape (tests/golden/make_golden_family_train.py --b24): fast_train_b24.npz in fp32 and fast_train_b24_f16.npz under
    torch.autocast("cpu", float16) + GradScaler (fast_fullsubnet/trainer.py:52-66).  Held: the fp32 step matches the fp32
    golden like the short sibling golden does; the 16-bit step is closer to the reference's fp32 step than the reference's OWN
    fp16-autocast step is, and within twice that distance of the fp16-autocast step itself."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.train import lstm2_train_chunks, train_step
    z32, meta = load(golden_dir, "fast_train_b24")
    z16, meta16 = load(golden_dir, "fast_train_b24_f16")
    assert meta16["autocast"] == "torch.float16" and meta16["batch"] == meta["batch"] == 24
    keys = [k[6:] for k in z32.files if k.startswith("gnorm/")]
    ref = (abs(float(z16["loss"]) - float(z32["loss"])) / float(z32["loss"]),
           abs(float(z16["total_norm"]) - float(z32["total_norm"])) / float(z32["total_norm"]),
           max(abs(float(z16["gnorm/" + k]) - float(z32["gnorm/" + k])) / float(z32["gnorm/" + k]) for k in keys))
    noisy = torch.from_numpy(O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_noisy"])).cuda()
    clean = torch.from_numpy((meta["clean_gain"] * O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_clean"]))
                             .astype(np.float32)).cuda()
    T = 1 + meta["length"] // 256 + FAST_KW["look_ahead"]
    Ts = fsn._lib.lib().fsn_fast_low_rate_frames(T, FAST_KW["shrink_size"])
    assert lstm2_train_chunks(Ts, meta["batch"] * FAST_KW["num_mels"], 12, 384) == (1536, 1)  # the path under test is taken
    out = {}
    for arith in ("f32", "f16"):
        m = _fast_train_model(fsn, meta, arith)
        opt = fsn.ClipAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
        scaler = torch.amp.GradScaler("cuda", enabled=arith != "f32")
        loss = train_step(m, opt, noisy, clean, scaler=scaler).item()
        assert opt.skipped_steps() == 0
        for tag, z in (("fp32", z32), ("fp16-autocast", z16)):
            out[arith, tag] = _step_distances(m, opt, loss, z, meta["sample"])
            d = out[arith, tag]
            print(f"{arith} step vs the reference's {tag} step: loss {d[0]:.2e}, total norm {d[1]:.2e}, worst tensor norm "
                  f"{d[2][1]:.2e} ({d[2][0]})")
    print(f"the reference's own fp16-autocast step vs its fp32 step: loss {ref[0]:.2e}, total norm {ref[1]:.2e}, worst tensor norm {ref[2]:.2e}")
    a = out["f32", "fp32"]
    assert a[0] <= 2e-6 and a[1] <= 5e-5 and a[2][1] <= 5e-4, a          # the bounds of the short sibling golden
    b = out["f16", "fp32"]
    assert b[1] <= ref[1] and b[2][1] <= ref[2], (b, ref)                  # closer to fp32 than the reference's fp16 step
    c = out["f16", "fp16-autocast"]
    assert c[1] <= 2 * ref[1] and c[2][1] <= 2 * ref[2], (c, ref)          # and within that distance of it


def test_trainer_trains_fast_fullsubnet_under_use_amp(fsn):
    """fast_fullsubnet/train_shrinkSize2.toml:5 (use_amp = true) through the Trainer mirror: the trainer's arithmetic reaches the
    bottleneck (24 utterances x 64 bands = one piece of 1536 rows on the 16-bit persistent training kernels), the reference's
    GradScaler stays at its initial scale, no step is skipped, the parameters move."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.trainer import Trainer
    m = _fast_train_model(fsn, dict(seed_w=5, gain=1.0), "f32")
    before = [p.detach().clone() for p in m.parameters()]
    loader = [(torch.from_numpy(O.make_noisy(24, 6144, seed=s)), torch.from_numpy(0.7 * O.make_noisy(24, 6144, seed=s + 9)))
              for s in (1, 2)]
    cfg = {"meta": {"use_amp": True}, "acoustics": {"n_fft": 512, "hop_length": 256, "win_length": 512, "sr": 16000},
           "trainer": {"train": {"epochs": 1, "clip_grad_norm_value": 10}}}
    opt = fsn.ClipAdam(m.parameters(), lr=1e-3)
    tr = Trainer(None, 0, cfg, False, False, m, None, opt, loader)
    assert tr.use_amp and tr.scaler.is_enabled() and tr._inner().train_arithmetic == "f16"
    tr._set_models_to_train_mode()
    loss = tr._train_epoch(1)
    assert np.isfinite(loss) and tr.scaler.get_scale() == 65536.0 and opt.skipped_steps() == 0
    assert m.bottleneck.train_arithmetic.startswith("f16")
    assert any(not torch.equal(p.detach(), q) for p, q in zip(m.parameters(), before))


@pytest.mark.parametrize("I,H,B,T", [(257, 512, 3, 23), (257, 512, 64, 9), (40, 384, 33, 12), (70, 320, 5, 8)])
def test_two_gru_layers_with_few_rows_on_the_chain_kernel(fsn, I, H, B, T):
    """nn.GRU(num_layers = 2) of a SequenceModel with few rows (sequence_model.py:59-66: the full-band model of a GRU FullSubNet)
    as ONE persistent launch - fb_chain_kernel with the GRU written as a four-gate cell r | z | nx | nh (fsn_gru2_forward) -
    against torch's nn.GRU on the CPU and against the library's own layer-by-layer path (gru_step_kernel), which it replaces
    where fsn_gru2_forward_supported says so (H = 384 / 512, up to 64 rows; H = 320 is no chain shape: layer by layer, never an error)."""
    from fullsubnet_amd.sequence_model import SequenceModel
    torch.manual_seed(7)
    m = SequenceModel(I, 0, H, 2, False, "GRU", None)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
    x = torch.randn(B, I, T)
    ref = torch.nn.GRU(I, H, 2, batch_first=True)
    ref.load_state_dict(m.sequence_model.state_dict())
    with torch.no_grad():
        want = ref(x.permute(0, 2, 1))[0].permute(0, 2, 1)  # [B, H, T]
        md = m.cuda().eval()
        L = fsn._lib.lib()
        Hp, Np = (H + 63) // 64 * 64, (B + 15) // 16 * 16
        on_chain = L.fsn_gru2_forward_supported(T, Np, Hp) == 1
        assert on_chain == (Hp in (384, 512)), "H = 384 / 512 with up to 64 rows: this device should hold the chain's grid"
        got = md(x.cuda()).cpu()
        # the layer-by-layer path of the same module
        from fullsubnet_amd import sequence_model as SM
        keep = SM.gru2_infer
        try:
            SM.gru2_infer = None
            layers, _ = md._inference_weights()
            h = SM.to_rows(x.cuda())
            for w_ih, w_hh, b_ih, b_hh in layers:
                h = SM.gru_layer_infer(h, w_ih, w_hh, b_ih, b_hh)
            steps = SM.from_rows(h[:, :, :H], B).cpu()
        finally:
            SM.gru2_infer = keep
    assert got.shape == want.shape == (B, H, T)
    d_ref, d_steps = (got - want).abs().max().item(), (got - steps).abs().max().item()
    print(f"GRU x 2, I = {I}, H = {H}, {B} rows, {T} steps: max |d| vs nn.GRU {d_ref:.2e}, vs the per-step kernels {d_steps:.2e}, "
          f"output range {want.min():.2f} .. {want.max():.2f}")
    assert float(want.abs().max()) > 0.3 and d_ref <= 2e-5 and d_steps <= 2e-5


def test_fullsubnet_gru_training_step_runs_and_learns(fsn):
    """The composed path under autograd: GRU forward-with-saves + BPTT through both blocks."""
    from oracle import fullsubnet_oracle as O
    from fullsubnet_amd.train import train_step
    kw = dict(num_freqs=257, look_ahead=2, sequence_model="GRU", fb_num_neighbors=0, sb_num_neighbors=15,
              fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
              sb_model_hidden_size=384, norm_type="offline_laplace_norm", num_groups_in_drop_band=2, weight_init=False)
    m = fsn.Model(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.make_params(seed=4, gates=3).items()}, strict=True)
    m = m.cuda().train()
    opt = fsn.ClipAdam(m.parameters(), lr=1e-3)
    noisy = torch.from_numpy(O.make_noisy(4, 4096, seed=1)).cuda()
    clean = torch.from_numpy(0.7 * O.make_noisy(4, 4096, seed=2)).cuda()
    losses = [train_step(m, opt, noisy, clean).item() for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[2] < losses[0]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_fullsubnet_other_hidden_sizes_run_composed(fsn):
    """Hidden sizes the fused kernels are not built for (sb != 384, fb not a multiple of 64) take the composed
    path; checked against the oracle."""
    from oracle import fullsubnet_oracle as O
    kw = dict(num_freqs=257, look_ahead=2, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
              fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=200,
              sb_model_hidden_size=96, norm_type="offline_laplace_norm", num_groups_in_drop_band=1, weight_init=False)
    params = O.make_params(seed=9, fb_hidden=200, sb_hidden=96, gain=2.0, mask_gain=8.0)
    m = fsn.Model(**kw)
    assert not m._fused
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    mag = np.abs(np.random.default_rng(1).standard_normal((2, 1, 257, 12))).astype(np.float32)
    with torch.no_grad():
        crm = m(torch.from_numpy(mag).cuda()).cpu().numpy()
    want = O.fullsubnet_forward(mag, params)
    assert np.abs(crm - want).max() <= 1e-4


@pytest.mark.parametrize("shape", [(3, 1, 257, 63), (2, 64, 12, 97), (2, 5, 33, 250), (1, 2, 7, 1)])
@pytest.mark.parametrize("name", ["offline_laplace_norm", "cumulative_laplace_norm", "offline_gaussian_norm",
                                  "cumulative_layer_norm", "forgetting_norm"])
def test_feature_norms_on_the_hip_kernels(fsn, name, shape):
    """norm_wrapper (audio_zen/model/base_model.py:356-372): fsn_norm (norm_kernels.hip) against the reference's tensor
    algebra as restated in base_model.py (itself pinned on the reference through the var_* goldens) - positive
    magnitude-like input, and for the zero-mean norms a signed one; 250 frames cross forgetting_norm's sample_length."""
    from fullsubnet_amd.base_model import BaseModel
    if name == "offline_gaussian_norm" and shape[1] * shape[2] * shape[3] < 2:
        pytest.skip("torch.std of one value")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(shape, generator=g) * 3.0 + 0.01
    if name in ("offline_gaussian_norm", "cumulative_layer_norm"):
        x = x - 1.2
    fn = getattr(BaseModel, name)
    want = fn(x.double()).float()                       # the algebra, in fp64 on the CPU
    algebra32 = fn(x.clone())                           # ... and in fp32 as the reference runs it
    got = fn(x.cuda())                                  # GPU tensor, no autograd: the HIP kernels
    assert got.is_cuda and got.shape == x.shape
    scale = want.abs().max().item()
    err = (got.cpu() - want).abs().max().item() / scale
    ref_err = (algebra32 - want).abs().max().item() / scale
    print(f"{name} {shape}: HIP vs fp64 algebra {err:.2e} (the reference's fp32 algebra: {ref_err:.2e})")
    assert err <= max(3 * ref_err, 2e-6)
    # and under autograd the same call is the algebra itself (training graphs): gradients flow
    xg = x.cuda().requires_grad_(True)
    fn(xg).sum().backward()
    assert xg.grad is not None and bool(torch.isfinite(xg.grad).all())


def test_improved_fullsubnet_offline_norm_of_a_five_dimensional_tensor(fsn):
    """improved_fullsubnet/model.py:124-216: the offline norm there runs over [B, N, 1, F_sub, T] with fp32 epsilon."""
    from fullsubnet_amd.improved_fullsubnet import BaseModel as IB
    x = torch.rand((2, 6, 1, 20, 40), generator=torch.Generator().manual_seed(5)) + 0.05
    want = IB.offline_laplace_norm(x.double()).float()
    got = IB.offline_laplace_norm(x.cuda())
    assert (got.cpu() - want).abs().max().item() <= 2e-6 * want.abs().max().item()


@pytest.mark.parametrize("I,ldx,rows", [(12, 48, 8304), (20, 32, 8304), (384, 384, 8304), (12, 16, 24688)])
def test_lstm_layer_with_left_over_tiles_on_the_persistent_kernels_vs_torch(fsn, I, ldx, rows):
    """fsn_lstm_layer_forward (inference) on row counts whose 16-row tiles do not divide into whole rounds (519 = 256 x 2 + 7,
    1543 = 512 x 3 + 7): the persistent kernels on the whole rounds, the left-over rows step by step beside them from their own
    small projection - input rows wider than the padded input (their compact copy goes step by step), a padded input width, the
    stacked form (I = H = ldx), several rounds.  Against ATen's nn.LSTM on the CPU (sequence_model.py:52-58), last rows included."""
    from fullsubnet_amd import sequence_model as SM
    torch.manual_seed(I + rows)
    T, H = 6, 384
    ref = torch.nn.LSTM(I, H, 1)
    x = torch.zeros(T, rows, ldx)
    x[:, :, :I] = torch.randn(T, rows, I)
    with torch.no_grad():
        want = ref(x[:, :, :I].contiguous())[0]
        p = [getattr(ref, n + "_l0").detach().cuda().contiguous() for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        got = SM.lstm_layer_infer(x.cuda(), *p).cpu()
    d = (got - want).abs()
    print(f"LSTM layer I = {I} (ldx {ldx}), {rows} rows: max |d| {d.max():.2e}, last 7 tiles {d[:, -112:].max():.2e}")
    assert got.shape == want.shape and d.max().item() <= 2e-5
