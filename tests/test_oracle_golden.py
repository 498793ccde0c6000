"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import ast
import os
import zlib

import numpy as np
import pytest

from oracle import fullsubnet_oracle as O


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = ast.literal_eval(str(z["meta"])) if "meta" in z else {}
    return z, meta


def inputs(meta):
    params = O.make_params(seed=meta["seed_w"], gain=meta["gain"], mask_gain=meta["mask_gain"])
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_x"])
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
    assert crc(noisy) == meta["crc_noisy"], "synthetic input generator drifted from the golden run"
    assert crc(np.concatenate([v.ravel() for v in params.values()])) == meta["crc_w"]
    return params, noisy


def ulp_at_frame_max(err, ref_re, ref_im):
    """|err| in units of the fp32 ULP of the per-frame max |X|."""
    fmax = np.maximum(np.abs(ref_re), np.abs(ref_im)).max(axis=1, keepdims=True)  # [B,1,T]
    ulp = np.spacing(fmax.astype(np.float32))
    return np.abs(err) / ulp


def test_elementwise_known_answers(golden_dir):
    z, _ = load(golden_dir, "elementwise")
    # decompress: clamp branch must be exact, log within a couple of ULP
    np.testing.assert_allclose(O.decompress_cIRM(z["m"]), z["dm"], rtol=3e-6, atol=2e-6)
    np.testing.assert_allclose(O.compress_cIRM(z["raw"]), z["comp"], rtol=3e-6, atol=1e-6)
    np.testing.assert_allclose(
        O.build_complex_ideal_ratio_mask(z["nr"], z["ni"], z["cr"], z["ci"]), z["cirm"], rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(O.drop_band(z["x"], 2), z["drop2"])
    np.testing.assert_array_equal(O.drop_band(z["x"], 3), z["drop3"])


@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd", "fsn_cumulative_b2"])
def test_stft_vs_reference(golden_dir, name):
    z, meta = load(golden_dir, name)
    _, noisy = inputs(meta)
    mag, _, re, im = O.stft(noisy, window=z["window"])
    assert re.shape == z["real"].shape
    # the oracle is the exactly-rounded DFT; MKL's fp32 FFT is within ~3 ULP of it at frame-max scale
    u = np.maximum(ulp_at_frame_max(re - z["real"], z["real"], z["imag"]),
                   ulp_at_frame_max(im - z["imag"], z["real"], z["imag"]))
    assert u.max() <= 2.0, u.max()  # measured 1.5
    assert np.percentile(u, 99) <= 1.5
    np.testing.assert_allclose(mag, z["mag"], rtol=0, atol=4 * np.spacing(np.float32(z["mag"].max())))


@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd", "fsn_cumulative_b2"])
def test_model_and_pipeline_vs_reference(golden_dir, name):
    z, meta = load(golden_dir, name)
    params, noisy = inputs(meta)
    kw = dict(norm_type=meta["norm_type"], num_groups_in_drop_band=meta["groups"])
    # stage 1: model on the REFERENCE's own magnitude -> isolates the network restatement
    crm, inter = O.fullsubnet_forward(z["mag"][:, None], params, return_intermediates=True, **kw)
    np.testing.assert_allclose(inter["fb_output"][:, 0], z["fb_output"], rtol=0, atol=2e-5)
    assert np.abs(crm - z["crm"]).max() <= 1e-4, np.abs(crm - z["crm"]).max()
    # stage 2: decompress + mask + istft on the reference's own crm
    dm = O.decompress_cIRM(z["crm"].transpose(0, 2, 3, 1))
    np.testing.assert_allclose(dm, z["dcrm"], rtol=1e-5, atol=1e-5)
    y = O.istft(z["enh_real"], z["enh_imag"], length=meta["length"], window=z["window"])
    scale = np.abs(z["enhanced"]).max()
    assert np.abs(y - z["enhanced"]).max() <= 2e-6 * scale
    # stage 3: end to end from the waveform
    y2 = O.full_band_crm_mask(noisy, params, window=z["window"], **kw)
    assert np.abs(y2 - z["enhanced"]).max() <= 2e-3 * scale  # decompress slope is ~100x near +-9.9


def test_config2_length_vs_reference(golden_dir):
    """The oracle at BASELINE config 2's sequence length (T = 188, 190 recurrent steps) against the reference's own
    output (tests/golden/fsn_long_b2.npz: every 4th bin / sample of 2 x 3 s utterances)."""
    z, meta = load(golden_dir, "fsn_long_b2")
    params, noisy = inputs(meta)
    b, ss = z["bins"], meta["sample_stride"]
    import torch
    win = torch.hann_window(512).numpy()  # the reference's window (SLEEF cosine; numpy's differs in the last bit)
    y, inter = O.full_band_crm_mask(noisy, params, window=win, return_intermediates=True)
    assert np.abs(inter["crm"][:, :, b] - z["crm"]).max() <= 1e-4
    assert np.abs(y[:, ::ss] - z["enhanced"]).max() <= 2e-3 * float(z["enhanced_absmax"])
    _, _, re, im = O.stft(noisy, window=win)
    ulp = np.spacing(z["frame_max"].astype(np.float32))  # per-frame max |X| over ALL bins of the reference
    u = np.maximum(np.abs(re[:, b] - z["real"]), np.abs(im[:, b] - z["imag"])) / ulp
    # the oracle is the exactly-rounded DFT; at 376 frames MKL's own error reaches 2.4 ULP here (2.95 in BASELINE.md)
    assert u.max() <= 3.0 and np.percentile(u, 99) <= 1.5, (u.max(), np.percentile(u, 99))


@pytest.mark.parametrize("name", ["fsn_offline_b2", "fsn_offline_b1_odd"])
def test_aten_baseline_reproduces_the_reference(golden_dir, name):
    """oracle/aten_baseline.py (what bench.py's cpu_baseline leg times) is the reference's own ATen operator sequence:
    it lands on the reference's golden outputs to rounding."""
    import torch
    from oracle import aten_baseline as A
    z, meta = load(golden_dir, name)
    params, noisy = inputs(meta)
    model = A.AtenFullSubNet(params).eval()
    y, crm = A.full_band_crm_mask(model, torch.from_numpy(noisy), return_crm=True)
    assert np.abs(crm.numpy() - z["crm"]).max() <= 2e-5
    assert np.abs(y.numpy() - z["enhanced"]).max() <= 1e-4 * np.abs(z["enhanced"]).max()


def test_dropband_eval_quirk(golden_dir):
    """Q1: the reference drops bands for any B > 1, eval mode included (model.py:114)."""
    z, meta = load(golden_dir, "fsn_dropband_b4")
    params, _ = inputs(meta)
    crm = O.fullsubnet_forward(z["mag"][:, None], params, norm_type=meta["norm_type"],
                               num_groups_in_drop_band=2)
    assert crm.shape == z["crm"].shape == (4, 2, 128, z["mag"].shape[-1])
    assert np.abs(crm - z["crm"]).max() <= 1e-4


def test_fp64_arbiter_agrees(golden_dir):
    z, meta = load(golden_dir, "fsn_offline_b2")
    params, _ = inputs(meta)
    c32 = O.fullsubnet_forward(z["mag"][:, None], params)
    c64 = O.fullsubnet_forward(z["mag"][:, None], params, dtype=np.float64)
    assert np.abs(c32 - c64).max() <= 1e-4
    assert np.abs(z["crm"] - c64).max() <= 1e-4


def test_freq_unfold_matches_docstring_shape():
    x = np.arange(2 * 1 * 161 * 5, dtype=np.float32).reshape(2, 1, 161, 5)
    u = O.freq_unfold(x, 9)
    assert u.shape == (2, 161, 1, 19, 5)  # base_model.py:23
    assert u[0, 0, 0, 9, 0] == x[0, 0, 0, 0] and u[0, 0, 0, 0, 0] == x[0, 0, 9, 0]
    assert u[1, 160, 0, 18, 3] == x[1, 0, 151, 3]
