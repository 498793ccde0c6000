"""Deterministic synthetic inputs and random weights (numpy PCG64: platform independent) shared by the
tests, the oracle, bench.py and tools/.  Generators only - not a compute path.  The golden fixtures under
tests/golden/ store CRCs of what these functions return: their output must never change."""
import numpy as np

FULLSUBNET_SHAPES = dict(num_freqs=257, fb_hidden=512, sb_hidden=384, sb_num_neighbors=15)


def make_params(seed=0, num_freqs=257, fb_hidden=512, sb_hidden=384, sb_num_neighbors=15,
                gain=1.0, mask_gain=1.0, dtype=np.float32, gates=4, fb_num_neighbors=0):
    """Random weights with the reference state_dict names/shapes.

    U(-1/sqrt(H), 1/sqrt(H)) like nn.LSTM / nn.Linear defaults, times ``gain``;
    the sub-band output layer (the compressed mask itself) is additionally
    scaled by ``mask_gain`` so that the mask spans +-10 and crosses the +-9.9
    clamp of decompress_cIRM - with default init it stays within +-0.07 and a
    1e-4 absolute check would be vacuous.
    """
    rng = np.random.default_rng(seed)
    p = {}

    def lstm(prefix, I, H):
        k = 1.0 / np.sqrt(H)
        for layer, isz in ((0, I), (1, H)):
            p[f"{prefix}.sequence_model.weight_ih_l{layer}"] = rng.uniform(-k, k, (gates * H, isz))
            p[f"{prefix}.sequence_model.weight_hh_l{layer}"] = rng.uniform(-k, k, (gates * H, H))
            p[f"{prefix}.sequence_model.bias_ih_l{layer}"] = rng.uniform(-k, k, (gates * H,))
            p[f"{prefix}.sequence_model.bias_hh_l{layer}"] = rng.uniform(-k, k, (gates * H,))

    def fc(prefix, I, O):
        k = 1.0 / np.sqrt(I)
        p[f"{prefix}.fc_output_layer.weight"] = rng.uniform(-k, k, (O, I))
        p[f"{prefix}.fc_output_layer.bias"] = rng.uniform(-k, k, (O,))

    lstm("fb_model", num_freqs, fb_hidden)
    fc("fb_model", fb_hidden, num_freqs)
    lstm("sb_model", (2 * sb_num_neighbors + 1) + (2 * fb_num_neighbors + 1), sb_hidden)
    fc("sb_model", sb_hidden, 2)
    for k in ("sb_model.fc_output_layer.weight", "sb_model.fc_output_layer.bias"):
        p[k] = p[k] * mask_gain
    return {k: (v * gain).astype(dtype) for k, v in p.items()}


def make_noisy(batch, length, seed=1234, dtype=np.float32):
    """Speech-like synthetic mix: 5 harmonics of f0 in U(100,300) Hz with
    4 Hz AM, plus white noise at an SNR drawn from [-5, 20] dB, scaled to ~ -26 dBFS."""
    rng = np.random.default_rng(seed)
    t = np.arange(length) / 16000.0
    out = np.empty((batch, length), dtype=np.float64)
    for b in range(batch):
        f0 = rng.uniform(100, 300)
        clean = sum(np.sin(2 * np.pi * f0 * (h + 1) * t + rng.uniform(0, 6.28)) / (h + 1) for h in range(5))
        clean *= 0.5 * (1 + np.sin(2 * np.pi * 4 * t + rng.uniform(0, 6.28)))
        noise = rng.standard_normal(length)
        snr = rng.uniform(-5, 20)
        noise *= np.sqrt((clean ** 2).mean() / ((noise ** 2).mean() * 10 ** (snr / 10)))
        mix = clean + noise
        out[b] = 0.05 * mix / np.sqrt((mix ** 2).mean())
    return out.astype(dtype)


# ---- sibling model families (Fast / Improved FullSubNet, full-band baseline) ---------------------
IMPROVED_16K = dict(n_fft=512, hop_length=128, win_length=512, fdrc=0.5, num_freqs=257, freq_cutoffs=[20, 80],
                    sb_num_center_freqs=[1, 4, 8], sb_num_neighbor_freqs=[15, 15, 15], fb_num_center_freqs=[1, 4, 8],
                    fb_num_neighbor_freqs=[15, 15, 15], fb_hidden_size=512, sb_hidden_size=384)
# the reference's own 48 kHz example (improved_fullsubnet/model.py:603-620)
IMPROVED_48K = dict(n_fft=960, hop_length=480, win_length=960, fdrc=0.5, num_freqs=481, freq_cutoffs=[20, 120, 240],
                    sb_num_center_freqs=[1, 4, 20, 60], sb_num_neighbor_freqs=[15, 15, 15, 15],
                    fb_num_center_freqs=[1, 4, 20, 60], fb_num_neighbor_freqs=[15, 15, 15, 15], fb_hidden_size=512,
                    sb_hidden_size=384)
# BASELINE config 5 names "769 bins" (n_fft 1536 at 48 kHz); the reference has no hyper-parameters for it (its own
# 48 kHz example is the 481-bin one above, improved_fullsubnet/model.py:603-620).  A scaling experiment with cut-offs
# chosen to satisfy (upper - lower) % centre == 0 (model.py:341-346): 32 + 40 + 12 + 6 = 90 units per utterance.
IMPROVED_48K_769 = dict(n_fft=1536, hop_length=768, win_length=1536, fdrc=0.5, num_freqs=769,
                        freq_cutoffs=[32, 192, 384], sb_num_center_freqs=[1, 4, 16, 64],
                        sb_num_neighbor_freqs=[15, 15, 15, 15], fb_num_center_freqs=[1, 4, 16, 64],
                        fb_num_neighbor_freqs=[15, 15, 15, 15], fb_hidden_size=512, sb_hidden_size=384)


def _block_params(rng, p, prefix, I, H, O_, num_layers):
    k = 1.0 / np.sqrt(H)
    for layer in range(num_layers):
        isz = I if layer == 0 else H
        p[f"{prefix}.sequence_model.weight_ih_l{layer}"] = rng.uniform(-k, k, (4 * H, isz))
        p[f"{prefix}.sequence_model.weight_hh_l{layer}"] = rng.uniform(-k, k, (4 * H, H))
        p[f"{prefix}.sequence_model.bias_ih_l{layer}"] = rng.uniform(-k, k, (4 * H,))
        p[f"{prefix}.sequence_model.bias_hh_l{layer}"] = rng.uniform(-k, k, (4 * H,))
    if O_:
        p[f"{prefix}.fc_output_layer.weight"] = rng.uniform(-k, k, (O_, H))
        p[f"{prefix}.fc_output_layer.bias"] = rng.uniform(-k, k, (O_,))


def make_fast_params(seed=0, gain=2.0, out_gain=8.0, num_mels=64, num_freqs=257, bottleneck_hidden=384, bottleneck_layers=2,
                     nn_noisy=5, nn_enc=0, dtype=np.float32):
    """Random weights with the reference state_dict names of fast_fullsubnet.model.Model (without
    ``mel_scale.fb``, which the tests take from the golden file / the product)."""
    rng = np.random.default_rng(seed)
    p = {}
    _block_params(rng, p, "encoder.0", num_mels, 384, 0, 1)
    _block_params(rng, p, "encoder.1", 384, 257, num_mels, 1)
    _block_params(rng, p, "bottleneck", (2 * nn_noisy + 1) + (2 * nn_enc + 1), bottleneck_hidden, 1, bottleneck_layers)
    _block_params(rng, p, "decoder_lstm.0", 2 * num_mels, 512, 0, 1)
    _block_params(rng, p, "decoder_lstm.1", 512, 512, 2 * num_freqs, 1)
    for k in ("decoder_lstm.1.fc_output_layer.weight", "decoder_lstm.1.fc_output_layer.bias"):
        p[k] = p[k] * out_gain  # the mask itself: spread it over a few units so that 1e-4 absolute means something
    return {k: (v * gain).astype(dtype) for k, v in p.items()}


def make_fullband_params(seed=0, gain=2.0, out_gain=8.0, num_freqs=257, hidden=512, dtype=np.float32):
    rng = np.random.default_rng(seed)
    p = {}
    _block_params(rng, p, "fullband_model", num_freqs, hidden, 2 * num_freqs, 3)
    for k in ("fullband_model.fc_output_layer.weight", "fullband_model.fc_output_layer.bias"):
        p[k] = p[k] * out_gain
    return {k: (v * gain).astype(dtype) for k, v in p.items()}



def make_improved_params(cfg, seed=0, gain=1.5, mask_gain=6.0, dtype=np.float32):
    """Random weights with the reference state_dict names of improved_fullsubnet.model.Model."""
    rng = np.random.default_rng(seed)
    p = {}
    F = cfg["num_freqs"] - 1
    _block_params(rng, p, "fb_model", F, cfg["fb_hidden_size"], F, 2)
    for i, (sc, sn, fc, fn) in enumerate(zip(cfg["sb_num_center_freqs"], cfg["sb_num_neighbor_freqs"],
                                             cfg["fb_num_center_freqs"], cfg["fb_num_neighbor_freqs"])):
        pre = f"sb_model.sb_models.{i}"
        _block_params(rng, p, pre, (sc + 2 * sn) + (fc + 2 * fn), cfg["sb_hidden_size"], 2 * sc, 2)
        for k in (f"{pre}.fc_output_layer.weight", f"{pre}.fc_output_layer.bias"):
            p[k] = p[k] * mask_gain
    return {k: (v * gain).astype(dtype) for k, v in p.items()}
