"""Parity of the HIP path (through the C ABI) against the reference's golden vectors and the CPU
oracle.  Needs a real MI355X:  python -m pytest tests -m gpu

Tolerances (north star): compressed cIRM within 1e-4 absolute (fp32); STFT bins within 2 ULP - asserted at
frame-max scale both against the exactly-rounded transform (<= 1 ULP) and against the reference's own MKL
output in the golden files (<= 2 ULP; measured 1.5).  At each bin's OWN scale no independent FFT can be within
2 ULP of MKL (cancellation bins); that figure is printed, not asserted."""
import ast
import os

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
    fullsubnet_amd._lib.lib()  # raises if libfsn_hip.so is missing: no fallback
    return fullsubnet_amd


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = ast.literal_eval(str(z["meta"])) if "meta" in z else {}
    return z, meta


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build_model(fsn, meta, groups=None, arith="f32"):
    params = O.make_params(seed=meta["seed_w"], gain=meta["gain"], mask_gain=meta["mask_gain"])
    m = fsn.Model(norm_type=meta["norm_type"], num_groups_in_drop_band=meta["groups"] if groups is None else groups,
                  **MODEL_KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m.arithmetic = arith  # "f32" unless a test asks for the opt-in split-precision kernels
    return m.cuda().eval(), params


ARITHS = ["f32", "f16x3"]


def ulp_at_frame_max(err, ref_re, ref_im):
    fmax = np.maximum(np.abs(ref_re), np.abs(ref_im)).max(axis=1, keepdims=True)
    return np.abs(err) / np.spacing(fmax.astype(np.float32))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd"])
def test_stft(fsn, golden_dir, name):
    z, meta = load(golden_dir, name)
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    mag, phase, re, im = fsn.stft(dev(noisy), 512, 256, 512)
    re, im, mag = re.cpu().numpy(), im.cpu().numpy(), mag.cpu().numpy()
    assert re.shape == z["real"].shape
    # vs the exactly-rounded transform (oracle, fp64 DFT of the fp32 frame*window product)
    omag, _, ore, oim = O.stft(noisy, window=z["window"])
    u = np.maximum(ulp_at_frame_max(re - ore, ore, oim), ulp_at_frame_max(im - oim, ore, oim))
    assert u.max() <= 1.0, f"vs fp64 truth: {u.max()} ULP"
    # vs the reference's MKL output
    u = np.maximum(ulp_at_frame_max(re - z["real"], z["real"], z["imag"]),
                   ulp_at_frame_max(im - z["imag"], z["real"], z["imag"]))
    assert u.max() <= 2.0 and np.percentile(u, 99) <= 1.5, (u.max(), np.percentile(u, 99))
    own = np.maximum(np.abs(re - z["real"]) / np.spacing(np.abs(z["real"])), np.abs(im - z["imag"]) / np.spacing(np.abs(z["imag"])))
    print(f"stft {name}: vs MKL max {u.max():.2f} ULP at frame-max scale, p99 {np.percentile(own, 99):.0f} ULP at own scale")
    np.testing.assert_allclose(mag, z["mag"], rtol=0, atol=4 * np.spacing(np.float32(z["mag"].max())))
    np.testing.assert_allclose(phase.cpu().numpy(), np.arctan2(im, re), atol=1e-6)


@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd"])
def test_istft(fsn, golden_dir, name):
    z, meta = load(golden_dir, name)
    y = fsn.istft((dev(z["enh_real"]), dev(z["enh_imag"])), 512, 256, 512, length=meta["length"],
                  input_type="real_imag").cpu().numpy()
    scale = np.abs(z["enhanced"]).max()
    assert np.abs(y - z["enhanced"]).max() <= 2e-6 * scale
    # complex input + default length, against the oracle
    c = torch.complex(dev(z["real"]), dev(z["imag"]))
    y2 = fsn.istft(c, 512, 256, 512).cpu().numpy()
    ref = O.istft(z["real"], z["imag"], window=z["window"])
    assert y2.shape == ref.shape
    assert np.abs(y2 - ref).max() <= 2e-6 * np.abs(ref).max()


def test_stft_istft_roundtrip_and_3d(fsn):
    y = dev(O.make_noisy(3, 7000, seed=5).reshape(1, 3, 7000))
    mag, _, re, im = fsn.stft(y, 512, 256, 512)
    assert mag.shape == (1, 3, 257, 28)
    back = fsn.istft((re[0], im[0]), 512, 256, 512, length=7000, input_type="real_imag")
    assert (back - y[0]).abs().max().item() <= 1e-6


def test_mask_algebra(fsn, golden_dir):
    z, _ = load(golden_dir, "elementwise")
    np.testing.assert_allclose(fsn.decompress_cIRM(dev(z["m"])).cpu().numpy(), z["dm"], rtol=3e-6, atol=2e-6)
    np.testing.assert_allclose(fsn.compress_cIRM(dev(z["raw"])).cpu().numpy(), z["comp"], rtol=3e-6, atol=1e-6)
    got = fsn.build_complex_ideal_ratio_mask(*(dev(z[k]) for k in ("nr", "ni", "cr", "ci"))).cpu().numpy()
    np.testing.assert_allclose(got, z["cirm"], rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(fsn.drop_band(dev(z["x"]), 2).cpu().numpy(), z["drop2"])
    np.testing.assert_array_equal(fsn.drop_band(dev(z["x"]), 3).cpu().numpy(), z["drop3"])


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd", "fsn_cumulative_b2", "fsn_dropband_b4"])
def test_model_forward_vs_reference(fsn, golden_dir, name, arith):
    z, meta = load(golden_dir, name)
    model, _ = build_model(fsn, meta, arith=arith)
    with torch.no_grad():
        crm = model(dev(z["mag"][:, None])).cpu().numpy()
    assert crm.shape == z["crm"].shape
    err = np.abs(crm - z["crm"])
    assert err.max() <= 1e-4, f"max |d cIRM| = {err.max():.3e} (L1 {err.mean():.3e})"
    assert np.abs(z["crm"]).max() > 5.0  # the check is not vacuous


@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd", "fsn_cumulative_b2", "fsn_long_b2"])
def test_fullband_stage_vs_reference(fsn, golden_dir, name):
    """Row A5 on its own: the intermediate ``fb_output`` (fullsubnet/model.py:95: look-ahead pad -> norm -> full-band
    LSTM x 2 -> Linear -> ReLU) of the reference against fsn_fullsubnet_fullband, incl. the 190-step utterances."""
    z, meta = load(golden_dir, name)
    model, _ = build_model(fsn, meta)
    if "mag" in z:
        mag = dev(z["mag"][:, None])
        want = z["fb_output"]
        bins = slice(None)
    else:  # long fixtures store a strided subset of the bins; the magnitude is recomputed from the seeded waveform
        noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
        mag = fsn.stft(dev(noisy), 512, 256, 512)[0].unsqueeze(1)
        want = z["fb_output"]
        bins = z["bins"]
    got = model.fullband_output(mag).cpu().numpy()[:, bins]
    assert got.shape == want.shape
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * max(scale, 1.0), (np.abs(got - want).max(), scale)
    assert scale > 0.1 and (want == 0).mean() > 0.05  # the ReLU is active and the stage is not trivially zero


@pytest.mark.parametrize("arith", ARITHS)
def test_config2_length_vs_reference(fsn, golden_dir, arith):
    """BASELINE config 2's sequence length pinned on the REFERENCE (not only on the oracle): 2 utterances x 48 000
    samples, T = 188 frames, 190 recurrent steps - STFT, compressed mask and enhanced waveform against
    tests/golden/fsn_long_b2.npz (fullsubnet/model.py:72-136, inferencer.py:130-145).  The fixture holds every 4th bin
    / sample.  Also run inside a 64-utterance batch, where these rows go through the persistent kernels."""
    z, meta = load(golden_dir, "fsn_long_b2")
    model, _ = build_model(fsn, meta, arith=arith)
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    b, ss = z["bins"], meta["sample_stride"]
    _, _, re, im = fsn.stft(dev(noisy), 512, 256, 512)
    re, im = re.cpu().numpy()[:, b], im.cpu().numpy()[:, b]
    ulp = np.spacing(z["frame_max"].astype(np.float32))  # per-frame max |X| over ALL bins of the reference
    u = np.maximum(np.abs(re - z["real"]), np.abs(im - z["imag"])) / ulp
    # 376 frames: MKL's own fp32 FFT is up to 2.95 ULP from the exact transform at this scale (BASELINE.md section 2:
    # median 0.44 / p99 1.36 / max 2.95), so against MKL the bound is MKL's error; vs the exact transform it is <= 1
    assert u.max() <= 3.0 and np.percentile(u, 99) <= 1.5, (u.max(), np.percentile(u, 99))
    _, _, ore, oim = O.stft(noisy, window=torch.hann_window(512).numpy())  # torch's window, like the product path
    assert (np.maximum(np.abs(re - ore[:, b]), np.abs(im - oim[:, b])) / ulp).max() <= 1.0
    enh, crm = model.enhance(dev(noisy), return_crm=True)
    err = np.abs(crm.cpu().numpy()[:, :, b] - z["crm"])
    print(f"config-2 length, {arith}: max |d cIRM| vs reference {err.max():.2e} (L1 {err.mean():.2e})")
    assert err.max() <= 1e-4, err.max()
    assert np.abs(z["crm"]).max() > 9.9  # the mask crosses the decompression clamp
    assert np.abs(enh.cpu().numpy()[:, ::ss] - z["enhanced"]).max() <= 2e-3 * float(z["enhanced_absmax"])
    # the same two utterances as rows of a config-2 sized batch (16 448 rows: persistent kernels + left-over tiles)
    big = np.concatenate([noisy, O.make_noisy(62, meta["length"], seed=99)], axis=0)
    _, crm64 = model.enhance(dev(big), return_crm=True)
    err64 = np.abs(crm64[:2].cpu().numpy()[:, :, b] - z["crm"])
    print(f"config-2 batch 64, {arith}: max |d cIRM| vs reference {err64.max():.2e}")
    assert err64.max() <= 1e-4, err64.max()


@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd", "fsn_cumulative_b2"])
def test_full_band_crm_mask_end_to_end(fsn, golden_dir, name):
    z, meta = load(golden_dir, name)
    model, params = build_model(fsn, meta)
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    enh, crm = model.enhance(dev(noisy), return_crm=True)
    crm = crm.cpu().numpy()
    assert np.abs(crm - z["crm"]).max() <= 1e-4
    scale = np.abs(z["enhanced"]).max()
    # the decompression slope reaches ~100 near |m| = 9.9, so 1e-4 on the mask is ~1e-2 relative there
    assert np.abs(enh.cpu().numpy() - z["enhanced"]).max() <= 2e-3 * scale
    # and the reference-shaped Inferencer (stft -> Model -> decompress -> mask -> istft) agrees with the fused call
    cfg = dict(inferencer=dict(type="full_band_crm_mask", args={}),
               acoustics=dict(n_fft=512, hop_length=256, win_length=512, sr=16000))
    inf = fsn.Inferencer(cfg, model=model)
    one = inf.full_band_crm_mask(dev(noisy[:1]), {})   # one utterance: ONE call of the library (fsn_enhance)
    assert one.shape == (meta["length"],)
    assert np.abs(one - z["enhanced"][0]).max() <= 2e-3 * scale
    inf.fused_call = False                             # stage by stage, as inferencer.py:130-145 spells it
    staged = inf.full_band_crm_mask(dev(noisy[:1]), {})
    assert np.abs(staged - z["enhanced"][0]).max() <= 2e-3 * scale
    assert np.abs(staged - one).max() <= 1e-6 * scale  # same arithmetic either way (measured: 0 - 2e-8 of the peak)


def test_inferencer_call_writes_the_reference_int16_files(fsn, golden_dir, tmp_path):
    """BaseInferencer.__call__ (audio_zen/inferencer/base_inferencer.py:163-195): every utterance of the loader through
    `full_band_crm_mask`, the enhanced waveform peak-normalised to 0.8 of int16 full scale and written as 16-bit PCM
    (:181-182), the noisy input beside it trimmed to the same length.  The file's samples are the reference's formula
    applied to the reference's own enhanced waveform (golden `fsn_offline_b2`) to within the one count per 2e-3 of
    peak the waveform bound allows; they are EXACTLY the formula applied to this path's waveform."""
    from scipy.io import wavfile
    z, meta = load(golden_dir, "fsn_offline_b2")
    model, _ = build_model(fsn, meta)
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    loader = [(torch.from_numpy(noisy[i:i + 1]), [f"utt{i}"]) for i in range(meta["batch"])]
    cfg = dict(inferencer=dict(type="full_band_crm_mask", args={}),
               acoustics=dict(n_fft=512, hop_length=256, win_length=512, sr=16000))
    inf = fsn.Inferencer(cfg, model=model, dataloader=loader, output_dir=str(tmp_path))
    inf()
    amp = np.iinfo(np.int16).max
    for i in range(meta["batch"]):
        sr, pcm = wavfile.read(str(inf.enhanced_dir / f"utt{i}.wav"))
        assert sr == 16000 and pcm.dtype == np.int16 and pcm.shape == (meta["length"],)
        own = inf.full_band_crm_mask(dev(noisy[i:i + 1]), {})
        assert np.array_equal(pcm, np.int16(0.8 * amp * own / np.max(np.abs(own))))
        ref = z["enhanced"][i]
        want = np.int16(0.8 * amp * ref / np.max(np.abs(ref)))
        assert np.abs(pcm.astype(np.int32) - want.astype(np.int32)).max() <= 1 + int(2 * 2e-3 * 0.8 * amp)
        assert abs(int(np.abs(pcm).max()) - int(0.8 * amp)) <= 1
        sr_n, noisy_file = wavfile.read(str(inf.noisy_dir / f"utt{i}.wav"))
        assert sr_n == 16000 and np.array_equal(noisy_file, noisy[i])
    with pytest.raises(AssertionError):  # base_inferencer.py:173: one utterance per loader item
        fsn.Inferencer(cfg, model=model, dataloader=[(torch.from_numpy(noisy), ["a", "b"])], output_dir=str(tmp_path))()


def test_batch_independence_and_determinism(fsn):
    """Every utterance of a batch gets the full mask, independent of its neighbours; two runs are
    bit-identical (no atomics on the path)."""
    meta = dict(seed_w=3, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, params = build_model(fsn, meta)
    noisy = O.make_noisy(5, 3000, seed=11)
    a = model.enhance(dev(noisy))
    b = model.enhance(dev(noisy))
    assert torch.equal(a, b)
    solo = model.enhance(dev(noisy[3:4]))
    assert (solo[0] - a[3]).abs().max().item() <= 1e-5 * a.abs().max().item()


def test_oracle_parity_fresh_weights(fsn):
    """Same seeded inputs through the HIP path and the CPU oracle (not via golden files)."""
    meta = dict(seed_w=7, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, params = build_model(fsn, meta)
    noisy = O.make_noisy(2, 2900, seed=21)
    enh, crm = model.enhance(dev(noisy), return_crm=True)
    ref, inter = O.full_band_crm_mask(noisy, params, window=torch.hann_window(512).numpy(), return_intermediates=True)
    assert np.abs(crm.cpu().numpy() - inter["crm"]).max() <= 1e-4
    assert np.abs(enh.cpu().numpy() - ref).max() <= 2e-3 * np.abs(ref).max()


@pytest.mark.parametrize("batch", [16, 24, 33, 48])
def test_more_row_tiles_than_cus(fsn, batch):
    """(Also the three shapes of the full-band chain kernel: one row tile with K split over four waves, two row tiles
    x two K halves at batch 24, one row tile per wave from batch 33.)
    B*F/16 > 256 tiles: the persistent recurrent kernel takes floor(tiles/CUs) tiles per CU and
    the left-over tiles run step by step on the auxiliary stream (B=16: 257 tiles -> 256 + 1;
    B=33: 531 tiles -> 2 x 256 + 19 -> general multi-round plan; B=48: 771 tiles -> 3 x 256 + 3).  At 2 - 4 tiles
    per workgroup the last layer forms its input projection itself (lstm_rec_x_kernel); at one tile per workgroup it
    runs on the projection GEMM + lstm_rec_kernel pair.  Every row must still match."""
    meta = dict(seed_w=5, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, params = build_model(fsn, meta)
    noisy = O.make_noisy(batch, 1300, seed=31)
    enh, crm = model.enhance(dev(noisy), return_crm=True)
    ref, inter = O.full_band_crm_mask(noisy, params, window=torch.hann_window(512).numpy(), return_intermediates=True)
    err = np.abs(crm.cpu().numpy() - inter["crm"])
    assert err.max() <= 1e-4, (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert np.abs(enh.cpu().numpy() - ref).max() <= 2e-3 * np.abs(ref).max()


@pytest.mark.parametrize("batch", [10, 24, 40, 96, 104, 128])
def test_full_rounds_plus_leftover_tiles_for_large_batches(fsn, batch):
    """More than five row tiles per CU: the plan takes FULL rounds of the persistent kernels and hands the few tiles that
    remain to the step kernels (96 utterances: 1542 tiles = two rounds of 3 tiles per workgroup + 6; 128: two rounds of 4
    + 8 - where whole rounds only would take three).  The model has no cross-utterance term: every utterance must equal
    its result in a batch of 32 (another plan: one round of 2 tiles per workgroup, held to the oracle above).  Likewise
    the batches BETWEEN the regimes (10, 24, 40: run as 8 + 2, 16 + 8, 32 + 8 by the cost model of run_core_chunks) against
    batches of 8 (the group kernel, held to the oracle by test_few_rows_on_the_group_kernel)."""
    meta = dict(seed_w=5, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, _ = build_model(fsn, meta)
    noisy = dev(O.make_noisy(batch, 1300, seed=37))
    enh, crm = model.enhance(noisy, return_crm=True)
    assert torch.isfinite(crm).all() and torch.isfinite(enh).all()
    piece = 32 if batch >= 96 else 8
    parts = [model.enhance(noisy[i:i + piece], return_crm=True) for i in range(0, batch, piece)]
    enh32, crm32 = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    assert (crm - crm32).abs().max().item() <= 2e-5
    assert (enh - enh32).abs().max().item() <= 1e-5 * enh32.abs().max().item()


@pytest.mark.parametrize("arith", ARITHS)
def test_config2_full_size(fsn, arith):
    """BASELINE config 2 at full size (batch 64 x 3 s): the oracle checks two utterances (a few
    seconds of CPU each); the rest is covered by size-independent properties - every utterance of
    the batch equals its own solo run (batch independence), and two runs are bit-identical."""
    meta = dict(seed_w=0, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, params = build_model(fsn, meta, arith=arith)
    noisy = O.make_noisy(64, 48000, seed=1234)
    x = dev(noisy)
    enh, crm = model.enhance(x, return_crm=True)
    enh2 = model.enhance(x)
    assert torch.equal(enh, enh2)
    assert crm.shape == (64, 2, 257, 188) and bool(torch.isfinite(enh).all())
    win = torch.hann_window(512).numpy()
    for b in (0, 63):
        ref, inter = O.full_band_crm_mask(noisy[b:b + 1], params, window=win, return_intermediates=True)
        err = np.abs(crm[b:b + 1].cpu().numpy() - inter["crm"])
        assert err.max() <= 1e-4, (b, err.max())
        assert np.abs(enh[b:b + 1].cpu().numpy() - ref).max() <= 2e-3 * np.abs(ref).max()
    for b in (17, 40):  # rows of utterance 63 / 17 / 40 live in different workgroups and the aux stream
        solo, solo_crm = model.enhance(x[b:b + 1], return_crm=True)
        assert (solo_crm[0] - crm[b]).abs().max().item() <= 5e-5
        assert (solo[0] - enh[b]).abs().max().item() <= 1e-3 * enh.abs().max().item()


def _rows_of(crm):
    """[B, 2, F, T] -> [B F, 2, T]: the row order fsn_fullsubnet_forward_rows uses (n = b F + f)."""
    B, _, F, T = crm.shape
    return crm.permute(0, 2, 1, 3).reshape(B * F, 2, T)


@pytest.mark.parametrize("name,cuts", [
    ("fsn_offline_b2", [0, 100, 257, 300, 514]),          # inside one utterance / aligned / across the boundary
    ("fsn_offline_b2", [0, 65, 129, 193, 257, 322, 386, 450, 514]),  # 8 ranks
    ("fsn_cumulative_b2", [0, 171, 343, 514]),            # per-row (cumulative) divisors indexed by global row
    ("fsn_offline_b1_odd", [0, 33, 65, 97, 129, 161, 193, 225, 257]),  # one utterance over 8 ranks
    ("fsn_dropband_b4", [0, 500, 1028]),
])
def test_row_slices_vs_reference(fsn, golden_dir, name, cuts):
    """the batch x frequency rows in contiguous slices (one per rank), each through
    fsn_fullsubnet_forward_rows; concatenated they are the reference's mask.  Slices cut utterances anywhere."""
    z, meta = load(golden_dir, name)
    model, _ = build_model(fsn, meta, groups=1)
    x = dev(z["mag"][:, None])
    with torch.no_grad():
        whole = _rows_of(model(x))
        parts = [model.forward_rows(x, lo, hi) for lo, hi in zip(cuts, cuts[1:])]
    got = torch.cat(parts, dim=0)
    assert got.shape == whole.shape
    assert (got - whole).abs().max().item() <= 2e-5  # other kernels for fewer rows, same arithmetic
    if meta["groups"] == 1:  # goldens with band dropping hold other rows; the unsharded forward is held to them
        err = np.abs(got.cpu().numpy() - _rows_of(torch.from_numpy(z["crm"])).numpy())
        assert err.max() <= 1e-4, err.max()
    # an utterance-aligned slice is the plain forward on those utterances, bit for bit
    F = 257
    if x.shape[0] >= 2:
        with torch.no_grad():
            a = model.forward_rows(x, F, 2 * F)
            b = _rows_of(model(x[1:2]))
        assert torch.equal(a, b)


def test_row_slice_on_the_persistent_kernel(fsn):
    """A slice large enough for the persistent recurrent kernel with left-over tiles beside it (258 row tiles:
    256 + 2) that starts and ends inside utterances, against the unsharded forward; and the workspace only covers
    the utterances the slice touches."""
    import ctypes
    meta = dict(seed_w=5, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, params = build_model(fsn, meta)
    noisy = O.make_noisy(20, 1300, seed=31)
    mag = fsn.stft(dev(noisy), 512, 256, 512)[0][:, None].contiguous()
    lo, hi = 300, 300 + 258 * 16 - 5
    with torch.no_grad():
        whole = _rows_of(model(mag))
        part = model.forward_rows(mag, lo, hi)
        tail = model.forward_rows(mag, hi, 20 * 257)
    assert (part - whole[lo:hi]).abs().max().item() <= 2e-5
    assert (tail - whole[hi:]).abs().max().item() <= 2e-5
    want = O.fullsubnet_forward(mag[2:3].cpu().numpy(), params)  # utterance 2 lies wholly inside the slice
    got = part[2 * 257 - lo:3 * 257 - lo].reshape(1, 257, 2, -1).permute(0, 2, 1, 3).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4
    L = fsn._lib.lib()
    T = mag.shape[-1]
    # the workspace depends on the utterances the slice touches, not on the batch around them
    assert (L.fsn_fullsubnet_rows_workspace_bytes(ctypes.byref(model._cfg), 20, T, lo, hi)
            == L.fsn_fullsubnet_rows_workspace_bytes(ctypes.byref(model._cfg), 50, T, lo, hi) > 0)
    with pytest.raises(fsn._lib.FsnError):
        model.forward_rows(mag, 10, 10)
    with pytest.raises(fsn._lib.FsnError):
        model.forward_rows(mag, 0, 20 * 257 + 1)


def test_enhance_row_sharded_single_process_equals_the_fused_call(fsn, golden_dir):
    """parallel.enhance_row_sharded without a process group (one rank owns every row): stft -> forward_rows ->
    decompress -> mask -> istft as separate calls must reproduce the fused fsn_enhance (the collective itself is
    covered on CPU, tests/test_parallel_cpu.py)."""
    from fullsubnet_amd.parallel import enhance_row_sharded
    z, meta = load(golden_dir, "fsn_offline_b2")
    model, _ = build_model(fsn, meta, groups=1)
    noisy = dev(O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"]))
    fused, crm = model.enhance(noisy, return_crm=True)
    split = enhance_row_sharded(model, noisy)
    assert split.shape == fused.shape
    assert (split - fused).abs().max().item() <= 1e-4 * fused.abs().max().item()  # same mask, other rounding points
    assert np.abs(split.cpu().numpy() - z["enhanced"]).max() <= 2e-3 * np.abs(z["enhanced"]).max()


def test_errors_are_loud(fsn):
    with pytest.raises(Exception):
        fsn.stft(torch.zeros(2, 4000), 512, 256, 512)  # CPU tensor: no fallback
    with pytest.raises(fsn._lib.FsnError):
        fsn.stft(torch.zeros(2, 4000).cuda(), 401, 100, 401)  # odd FFT size -> error code from the C ABI
    with pytest.raises(fsn._lib.FsnError):
        fsn.stft(torch.zeros(2, 100).cuda(), 512, 256, 512)  # shorter than the reflect padding
    with pytest.raises(NotImplementedError):
        fsn.Model(norm_type="no_such_norm", num_groups_in_drop_band=1, **MODEL_KW)
    m = fsn.Model(norm_type="offline_laplace_norm", num_groups_in_drop_band=1, **MODEL_KW).cuda().eval()
    with pytest.raises(fsn._lib.FsnError):
        m.enhance(torch.zeros(1, 4000).cuda(), n_fft=1024, hop_length=512)  # the fused path is 512 / 256 only


@pytest.mark.parametrize("F,la,nb,fbh,B,T,norm", [
    (257, 0, 15, 512, 1, 1, "offline_laplace_norm"),      # a single frame, no look-ahead
    (161, 1, 7, 256, 3, 9, "offline_laplace_norm"),       # another spectrum size / neighbourhood / full-band width
    (129, 3, 0, 128, 2, 6, "cumulative_laplace_norm"),    # no neighbours at all: K = 2 -> one padded chunk
    (65, 2, 31, 64, 5, 4, "cumulative_laplace_norm"),     # neighbourhood almost as wide as the spectrum
    (257, 2, 15, 512, 17, 3, "offline_laplace_norm"),     # 273 row tiles: persistent kernel + 17 left-over tiles
    (257, 2, 15, 512, 8, 3, "offline_laplace_norm"),      # 129 row tiles: 32 groups of four through the
                                                          # one-workgroup-per-CU step kernel + 1 tile beside it
    (161, 1, 7, 256, 11, 4, "cumulative_laplace_norm"),   # 111 row tiles: 27 groups (two unit groups per wave) + 3
    (257, 1, 15, 512, 6, 2, "offline_laplace_norm"),      # 97 row tiles: the first size of the step regime
])
def test_fused_forward_other_shapes_vs_oracle(fsn, F, la, nb, fbh, B, T, norm):
    """fsn_fullsubnet_forward away from the shipped configuration (cfg fields are free: num_freqs, look_ahead,
    sb_num_neighbors, fb_hidden) against the oracle."""
    params = O.make_params(seed=F + nb, num_freqs=F, fb_hidden=fbh, sb_hidden=384, sb_num_neighbors=nb, gain=2.0,
                           mask_gain=12.0)
    kw = dict(MODEL_KW, num_freqs=F, look_ahead=la, sb_num_neighbors=nb, fb_model_hidden_size=fbh)
    m = fsn.Model(norm_type=norm, num_groups_in_drop_band=1, **kw)
    assert m._fused
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    mag = (np.abs(np.random.default_rng(B * T).standard_normal((B, 1, F, T))) + 0.05).astype(np.float32)
    with torch.no_grad():
        crm = m(dev(mag)).cpu().numpy()
    want = O.fullsubnet_forward(mag, params, look_ahead=la, sb_num_neighbors=nb, norm_type=norm)
    assert crm.shape == want.shape == (B, 2, F, T)
    assert np.abs(crm - want).max() <= 1e-4  # the north-star bound on the compressed mask, unscaled


@pytest.mark.parametrize("B,L", [(2, 300), (1, 257), (3, 1024), (2, 4097)])
def test_enhance_short_and_ragged_lengths_vs_oracle(fsn, B, L):
    """fsn_enhance at the shortest legal inputs (L = n_fft / 2 + 1: two frames, all reflect padding) and lengths
    that are / are not multiples of the hop, against the oracle's full path."""
    params = O.make_params(seed=1, gain=2.0, mask_gain=24.0)
    m = fsn.Model(norm_type="offline_laplace_norm", num_groups_in_drop_band=1, **MODEL_KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    noisy = O.make_noisy(B, L, seed=L)
    enh, crm = m.enhance(dev(noisy), return_crm=True)
    want, inter = O.full_band_crm_mask(noisy, params, return_intermediates=True)
    assert enh.shape == (B, L) and crm.shape == inter["crm"].shape
    assert np.abs(crm.cpu().numpy() - inter["crm"]).max() <= 1e-4
    assert np.abs(enh.cpu().numpy() - want).max() <= 2e-3 * max(np.abs(want).max(), 1e-6)


def test_poisoned_buffers_do_not_leak_into_results(fsn, monkeypatch):
    """Every output element is written and no kernel reads workspace it has not written: with all output
    tensors and the whole workspace pre-filled with NaN bit patterns the results are finite and bit-identical
    to a normal run (padded rows / bins / look-ahead frames included)."""
    params = O.make_params(seed=0, gain=2.0, mask_gain=24.0)
    noisy = dev(O.make_noisy(5, 3000, seed=3))  # 81 row tiles: step path; 17 utterances below: persistent + left-over
    noisy17 = dev(O.make_noisy(17, 1500, seed=4))

    def run(model):
        out = []
        for y in (noisy, noisy17):
            enh, crm = model.enhance(y, return_crm=True)
            mag, _, re, im = fsn.stft(y, 512, 256, 512)
            with torch.no_grad():
                fwd = model(mag.unsqueeze(1))
            back = fsn.istft((re, im), 512, 256, 512, length=y.shape[1], input_type="real_imag")
            out += [enh, crm, mag, re, im, fwd, back]
        torch.cuda.synchronize()
        return [t.clone() for t in out]

    for norm in ("offline_laplace_norm", "cumulative_laplace_norm"):
        m = fsn.Model(norm_type=norm, num_groups_in_drop_band=1, **MODEL_KW)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        m = m.cuda().eval()
        clean = run(m)
        real_empty, real_ws = torch.empty, fsn._lib.workspace

        def poisoned_empty(*a, **k):
            t = real_empty(*a, **k)
            if t.is_cuda and t.dtype == torch.float32:
                t.fill_(float("nan"))
            return t

        def poisoned_ws(nbytes, device):
            return real_ws(nbytes, device).fill_(0xFF)  # 0xFFFFFFFF is a NaN

        monkeypatch.setattr(torch, "empty", poisoned_empty)
        monkeypatch.setattr(fsn._lib, "workspace", poisoned_ws)
        try:
            dirty = run(m)
        finally:
            monkeypatch.undo()
        for a, b in zip(clean, dirty):
            assert torch.isfinite(b).all()
            assert torch.equal(a, b)


def test_fused_forward_random_configurations(fsn):
    """Property-style sweep (fixed seed): 24 random (num_freqs, look_ahead, neighbours, full-band width, batch,
    frames, norm) combinations of the fused forward against the oracle - step path, wavefront path and the
    persistent kernel with left-over tiles all occur."""
    rng = np.random.default_rng(12345)
    worst = 0.0
    for it in range(24):
        F = int(rng.choice([65, 129, 161, 257]))
        la = int(rng.integers(0, 4))
        nb = min(int(rng.choice([0, 3, 7, 15])), F - 1)
        fbh = int(rng.choice([64, 128, 256, 512]))
        B, T = int(rng.integers(1, 21)), int(rng.integers(1, 13))
        norm = str(rng.choice(["offline_laplace_norm", "cumulative_laplace_norm"]))
        params = O.make_params(seed=it, num_freqs=F, fb_hidden=fbh, sb_hidden=384, sb_num_neighbors=nb, gain=2.0,
                               mask_gain=12.0)
        kw = dict(MODEL_KW, num_freqs=F, look_ahead=la, sb_num_neighbors=nb, fb_model_hidden_size=fbh)
        m = fsn.Model(norm_type=norm, num_groups_in_drop_band=1, **kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        m = m.cuda().eval()
        mag = (np.abs(rng.standard_normal((B, 1, F, T))) + 0.05).astype(np.float32)
        with torch.no_grad():
            crm = m(dev(mag)).cpu().numpy()
        want = O.fullsubnet_forward(mag, params, look_ahead=la, sb_num_neighbors=nb, norm_type=norm)
        lim = 1e-4  # the north-star bound on the compressed mask, unscaled
        err = np.abs(crm - want).max()
        assert err <= lim, (F, la, nb, fbh, B, T, norm, err)
        worst = max(worst, err / lim)
    assert worst <= 0.5  # measured 0.07: a drift towards the limit is worth a look before it becomes a failure


def _tone_burst(batch, samples):
    """One 30 ms tone burst per utterance in digital silence: after the utterance-level norm the few bins it
    occupies are thousands of times the mean - the widest dynamic range the sub-band input can have."""
    x = np.zeros((batch, samples), np.float32)
    n = np.arange(480)
    for b in range(batch):
        x[b, 4000 + 100 * b:4480 + 100 * b] = 0.8 * np.sin(2 * np.pi * (1000 + 250 * b) / 16000 * n) * np.hanning(480)
    return x


def test_experimental_f16x3_matches_fp32(fsn):
    """The opt-in split-precision kernels (Model.arithmetic = "f16x3" -> cfg.arith; both sub-band recurrent layers and
    the projection between them): same mask as the fp32 path to well inside the parity budget, on the usual noisy
    input and on tone bursts in silence (fp16 range of the staged layer-0 input)."""
    params = O.make_params(seed=0, gain=2.0, mask_gain=24.0)
    m = fsn.Model(norm_type="offline_laplace_norm", num_groups_in_drop_band=1, **MODEL_KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    noisy, burst = dev(O.make_noisy(16, 2048, seed=1)), dev(_tone_burst(16, 16000))  # 257 row tiles each
    with pytest.raises(fsn._lib.FsnError):
        m.arithmetic = "fp8"
    assert m.arithmetic == "f32"
    ref, ref_b = m.enhance(noisy, return_crm=True)[1].cpu().numpy(), m.enhance(burst, return_crm=True)[1].cpu().numpy()
    m.arithmetic = "f16x3"
    got, got_b = m.enhance(noisy, return_crm=True)[1].cpu().numpy(), m.enhance(burst, return_crm=True)[1].cpu().numpy()
    m.arithmetic = "f32"
    assert np.array_equal(m.enhance(noisy, return_crm=True)[1].cpu().numpy(), ref)  # and back, bit for bit
    assert not np.array_equal(got, ref)                   # the switch really changed the arithmetic ...
    dev_noisy = np.abs(got - ref).max()
    assert dev_noisy <= 2e-5                              # ... and stayed within a fifth of the 1e-4 budget
    assert np.isfinite(got_b).all() and np.isfinite(ref_b).all()
    dev_burst = np.abs(got_b - ref_b).max()
    print(f"f16x3 vs fp32 mask deviation: noisy {dev_noisy:.2e}, tone burst {dev_burst:.2e}")
    assert dev_burst <= 1e-4


def test_f16x3_promotion_criterion(fsn):
    """The round-3 verdict's rule for the opt-in split-precision arithmetic (Model.arithmetic = "f16x3"): against the
    fp64 oracle ON THE SAME MAGNITUDES its error may be at most 2x the fp32 path's - maximum and rms - on inputs chosen
    to hurt: a noisy batch at the BASELINE length, tone bursts in digital silence (the widest dynamic range the
    normalised sub-band input can have), long utterances (1000 recurrent steps here; FSN_F16X3_FULL=1: 4000, 64 s - the
    measured run is profiles/r04_f16x3_promotion.txt: ratios 1.26 - 1.86 / 1.00 - 1.06), weights that drive the
    compressed mask beyond the +-9.9 clamp.  Batches of 16 = 257 row tiles, the smallest plan the split kernels run on."""
    params = O.make_params(seed=0, gain=2.0, mask_gain=24.0)
    m = fsn.Model(norm_type="offline_laplace_norm", num_groups_in_drop_band=1, **MODEL_KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    steps = 4000 if os.environ.get("FSN_F16X3_FULL") else 1000
    cases = {"noisy 16 x 3 s": (O.make_noisy(16, 48000, seed=5), [0, 15]), "tone bursts 16 x 1 s": (_tone_burst(16, 16000), [0, 15]),
             f"16 utterances of {steps} steps": (O.make_noisy(16, steps * 256, seed=6), [3])}
    for name, (x, rows) in cases.items():
        xd = dev(x)
        mag = fsn.stft(xd[rows], 512, 256, 512)[0].cpu().numpy()
        want = np.concatenate([O.fullsubnet_forward(mag[b:b + 1, None], params, dtype=np.float64) for b in range(len(rows))])
        err, full32 = {}, None
        for arith in ("f32", "f16x3"):
            m.arithmetic = arith
            full = m.enhance(xd, return_crm=True)[1]
            if arith == "f32":
                full32 = full
            else:
                assert not torch.equal(full, full32), "the split-precision kernels did not run on this plan"
            d = full[rows].cpu().numpy().astype(np.float64) - want
            err[arith] = (float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())))
        m.arithmetic = "f32"
        r_max, r_rms = err["f16x3"][0] / err["f32"][0], err["f16x3"][1] / err["f32"][1]
        print(f"{name}: fp32 max {err['f32'][0]:.2e} rms {err['f32'][1]:.2e} | f16x3 max {err['f16x3'][0]:.2e} rms "
              f"{err['f16x3'][1]:.2e} | ratios {r_max:.2f} / {r_rms:.2f} | mask {want.min():.1f} .. {want.max():.1f}")
        assert np.abs(want).max() > 9.9          # the clamp region is exercised
        assert err["f32"][0] <= 1e-4 and err["f16x3"][0] <= 1e-4   # both inside the north-star bound against the fp64 truth
        assert r_max <= 2.0 and r_rms <= 2.0, (name, r_max, r_rms)


def test_long_utterance_mask_bound_of_the_fp32_path(fsn):
    """The gate non-linearities of the forward kernels are hardware-transcendental forms (v_exp_f32 / v_rcp_f32,
    fsn_common.h; SURVEY 7 advises libm): their ~1 ULP errors feed back through the recurrence, so the 1e-4 bound on the
    compressed mask is checked where that has had the longest to act - 4000 recurrent steps (64 s of audio, 20x BASELINE's
    length), a batch of 32 (8224 sub-band rows: the persistent pair lstm_rec_in / lstm_rec_x at two row tiles per
    workgroup, the kernels the headline is measured on), weights that drive the mask beyond the +-9.9 clamp - against the
    fp64 oracle of one utterance from the middle of the batch."""
    steps = 4000
    params = O.make_params(seed=0, gain=2.0, mask_gain=24.0)
    m = fsn.Model(norm_type="offline_laplace_norm", num_groups_in_drop_band=1, **MODEL_KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m = m.cuda().eval()
    plan = fsn._lib.core_plan(m._cfg, 32, 1 + steps)
    assert plan["persistent_workgroups"] > 0 and plan["chunks"] == 1, plan
    x = O.make_noisy(32, steps * 256, seed=6)
    xd = dev(x)
    row = 17
    mag = fsn.stft(xd[row:row + 1], 512, 256, 512)[0].cpu().numpy()
    # the oracle's pieces in fullsubnet_forward's order (fullsubnet/model.py:85-135), fp64; the sub-band model on 40 of
    # the 257 independent rows (both edges with their reflected windows, and a spread over the spectrum)
    f64 = np.float64
    xp = np.pad(mag[:, None].astype(f64), [(0, 0), (0, 0), (0, 0), (0, 2)])
    fb_out = O.sequence_model(O.offline_laplace_norm(xp, dtype=f64).reshape(1, 257, -1), params, "fb_model",
                              activation="ReLU", dtype=f64).reshape(1, 1, 257, -1)
    sb_in = np.concatenate([O.freq_unfold(xp, 15).reshape(1, 257, 31, -1), O.freq_unfold(fb_out, 0).reshape(1, 257, 1, -1)], axis=2)
    sb_in = O.offline_laplace_norm(sb_in, dtype=f64)[0]
    bins = sorted(set(list(range(0, 8)) + list(range(249, 257)) + list(range(8, 249, 10))))
    want = O.sequence_model(np.ascontiguousarray(sb_in[bins]), params, "sb_model", activation=None, dtype=f64)[:, :, 2:]
    want = want.transpose(1, 0, 2)[None]  # [1, 2, bins, T]
    crm = m.enhance(xd, return_crm=True)[1][row:row + 1][:, :, bins].cpu().numpy().astype(np.float64)
    d = np.abs(crm - want)
    late = d[..., steps // 2:].max()
    print(f"{steps} steps, batch 32 (plan {plan}): max |d cIRM| vs the fp64 oracle {d.max():.2e} (second half of the "
          f"utterance {late:.2e}), rms {np.sqrt((d ** 2).mean()):.2e}, mask {want.min():.1f} .. {want.max():.1f}")
    assert np.abs(want).max() > 9.9
    assert d.max() <= 1e-4, d.max()
    del xd
    torch.cuda.empty_cache()


def test_two_streams_and_two_host_threads_are_independent(fsn):
    """re-entrant across streams.  The library's only state is one record per (device, caller stream)
    (auxiliary stream + fork / join events for the left-over tiles, profiler events): two host threads, each on its
    own torch stream, run the 257-row-tile plan (persistent kernel + a left-over tile on the auxiliary stream)
    concurrently and repeatedly; every result is bit-identical to the serial one."""
    import threading
    meta = dict(seed_w=0, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, _ = build_model(fsn, meta)
    inputs = [dev(O.make_noisy(16, 2048, seed=s)) for s in (1, 2, 3)]
    serial = [model.enhance(x).clone() for x in inputs[:2]]
    model.packed_weights()
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                out = None
                for _ in range(3):
                    out = model.enhance(inputs[i])
                st.synchronize()
                results[i] = out.clone()
        except Exception as e:  # surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for got, want in zip(results, serial):
        assert torch.equal(got, want)
    # profiler records are per stream as well: a profiled call on a side stream does not disturb the default stream's
    # and the switch itself is per stream (fsn_profile_enable(stream, on)): a stream that did not ask records nothing
    side, quiet = torch.cuda.Stream(), torch.cuda.Stream()
    fsn._lib.profile_enable(True, inputs[0].device)
    try:
        model.enhance(inputs[0])
        main_ms = fsn._lib.profile_read(inputs[0].device)
        with torch.cuda.stream(side):
            fsn._lib.profile_enable(True, inputs[1].device)
            model.enhance(inputs[1])
            side_ms = fsn._lib.profile_read(inputs[1].device)
            fsn._lib.profile_enable(False, inputs[1].device)
        with torch.cuda.stream(quiet):
            model.enhance(inputs[2])
            quiet_ms = fsn._lib.profile_read(inputs[2].device)
        again = fsn._lib.profile_read(inputs[0].device)
    finally:
        fsn._lib.profile_enable(False, inputs[0].device)
    assert main_ms["sb_rec_l0"] > 0 and side_ms["sb_rec_l0"] > 0
    assert all(v == 0 for v in quiet_ms.values()), quiet_ms
    # the same events read twice (elapsed times are re-derived from the timestamps: equal to rounding)
    assert again.keys() == main_ms.keys()
    assert all(again[k] == pytest.approx(main_ms[k], rel=1e-3, abs=1e-5) for k in main_ms)


def test_minute_long_utterances_chain_kernel_vs_per_step_launches(fsn):
    """The full-band chain kernel addresses its per-step hand-off buffers through 2 GB buffer resources: 4095 steps;
    longer inputs fall back to the wavefront of per-step launches.  Both paths on the same audio - 70 s (4376 frames,
    wavefront) and its first 60 s (3751 frames, chain kernel) - with the causal cumulative norm: the first 59 s agree,
    which also holds the chain kernel to the per-step kernels over 3700 dependent steps."""
    meta = dict(seed_w=0, gain=2.0, mask_gain=24.0, norm_type="cumulative_laplace_norm", groups=1)
    model, _ = build_model(fsn, meta)
    x = dev(O.make_noisy(2, 16000 * 70, seed=3))
    long = model.enhance(x)
    short = model.enhance(x[:, :16000 * 60].contiguous())
    assert bool(torch.isfinite(long).all())
    assert (long[:, :16000 * 59] - short[:, :16000 * 59]).abs().max().item() <= 2e-5 * short.abs().max().item()


def test_a_spin_bound_poisons_the_output(fsn):
    """The persistent kernels bound every spin; a launch that hit a bound raises its status word and the follow-up
    kernel (poison_if_kernel, here through its test hook) turns the output into NaN - never silent garbage - and
    leaves it alone otherwise."""
    L = fsn._lib.lib()
    out = torch.arange(5000, dtype=torch.float32, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    stream = fsn._lib.stream_ptr(out.device)
    fsn._lib.check(L.fsn_debug_poison_if(status.data_ptr(), fsn._lib.dev_ptr(out), out.numel(), stream))
    assert torch.equal(out, torch.arange(5000, dtype=torch.float32, device="cuda"))
    assert fsn._lib.stream_status(out.device) == (0, 0)
    status.fill_(7)
    fsn._lib.check(L.fsn_debug_poison_if(status.data_ptr(), fsn._lib.dev_ptr(out), out.numel(), stream))
    assert bool(torch.isnan(out).all())
    # ... and the host hears of it: the stream's sticky status carries the status word until it is cleared
    try:
        with pytest.raises(fsn._lib.FsnTimeout):
            fsn._lib.stream_status(out.device)
        assert fsn._lib.stream_status(out.device, raise_on_timeout=False) == (7, 1)
    finally:
        fsn._lib.stream_status_clear(out.device)
    assert fsn._lib.stream_status(out.device) == (0, 0)


@pytest.mark.parametrize("batch", [6, 8, 9])
def test_few_rows_on_the_group_kernel(fsn, batch):
    """6 - 9 utterances (97 - 145 row tiles, the per-rank share of a strong-scaled batch): both sub-band layers and
    the output layer run as ONE persistent launch in which clusters of eight workgroups exchange hidden-state slices
    through global memory (lstm_group_kernels.hip); rows that do not fill a cluster run beside it step by step
    (batch 6: 24 clusters + 1 tile, batch 8: 32 + 1, batch 9: 32 + 17).  Against the oracle, against every
    utterance's solo run (other kernels), and bit-identical from run to run."""
    meta = dict(seed_w=0, gain=2.0, mask_gain=24.0, norm_type="offline_laplace_norm", groups=1)
    model, params = build_model(fsn, meta)
    noisy = O.make_noisy(batch, 12000, seed=400 + batch)  # 47 frames -> 49 steps
    x = dev(noisy)
    enh, crm = model.enhance(x, return_crm=True)
    for _ in range(3):
        enh2, crm2 = model.enhance(x, return_crm=True)
        assert torch.equal(crm, crm2) and torch.equal(enh, enh2)
    assert bool(torch.isfinite(crm).all())
    win = torch.hann_window(512).numpy()
    for b in (0, batch - 1):
        ref, inter = O.full_band_crm_mask(noisy[b:b + 1], params, window=win, return_intermediates=True)
        err = np.abs(crm[b:b + 1].cpu().numpy() - inter["crm"])
        assert err.max() <= 1e-4, (b, err.max())
        assert np.abs(enh[b:b + 1].cpu().numpy() - ref).max() <= 2e-3 * np.abs(ref).max()
    for b in range(batch):
        _, solo = model.enhance(x[b:b + 1], return_crm=True)
        assert (solo[0] - crm[b]).abs().max().item() <= 5e-5, b
    # the cumulative norm (per-row, per-step divisors) through the same kernel
    meta_c = dict(meta, norm_type="cumulative_laplace_norm")
    model_c, params_c = build_model(fsn, meta_c)
    _, crm_c = model_c.enhance(x, return_crm=True)
    _, inter = O.full_band_crm_mask(noisy[:1], params_c, window=win, return_intermediates=True,
                                    norm_type="cumulative_laplace_norm")
    assert np.abs(crm_c[:1].cpu().numpy() - inter["crm"]).max() <= 1e-4
