"""Pin the training-step oracle (explicit LSTM cell + autograd) against the golden vectors produced
by the reference's own training step (tests/golden/make_golden_train.py).  CPU only."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import fullsubnet_oracle as O
from oracle import train_oracle as TO


def load(golden_dir, name="fsn_train_b4"):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return z, ast.literal_eval(str(z["meta"]))


def test_restated_lstm_cell_matches_aten():
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(7, 64, num_layers=1, batch_first=True)
    x = torch.randn(5, 9, 7, requires_grad=True)
    ref, _ = lstm(x)
    ref.square().sum().backward()
    gref = [x.grad.clone()] + [p.grad.clone() for p in lstm.parameters()]
    x2 = x.detach().clone().requires_grad_(True)
    ps = [p.detach().clone().requires_grad_(True) for p in lstm.parameters()]
    out = TO.lstm_layer(x2, *ps)
    out.square().sum().backward()
    assert torch.allclose(out, ref, atol=1e-6)
    for a, b in zip([x2.grad] + [p.grad for p in ps], gref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["fsn_train_b4", "fsn_train_cum_b4"])
def test_train_step_oracle_vs_reference(golden_dir, name):
    """offline_laplace_norm (fullsubnet/train.toml:82) and cumulative_laplace_norm (train_cumulativeLaplaceNorm.toml:82)."""
    z, meta = load(golden_dir, name)
    params = O.make_params(seed=meta["seed_w"])
    noisy = O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_noisy"])
    clean = (meta["clean_gain"] * O.make_noisy(meta["batch"], meta["length"], seed=meta["seed_clean"])).astype(np.float32)
    r = TO.train_step(params, noisy, clean, groups=meta["groups"], norm_type=meta.get("norm_type", "offline_laplace_norm"))
    assert abs(r["loss"] - float(z["loss"])) <= 1e-5 * float(z["loss"])
    s = meta["sample"]
    for k in params:
        g = r["grads"][k].reshape(-1)[::s].numpy()
        gn = float(z["gnorm/" + k])
        assert np.abs(g - z["g/" + k]).max() <= 2e-3 * max(np.abs(z["g/" + k]).max(), 1e-3 * gn) + 1e-9, k
        assert abs(float(r["grads"][k].norm()) - gn) <= 2e-3 * gn + 1e-9, k
        p = r["new_params"][k].reshape(-1)[::s].numpy()
        # Adam's first step moves every weight by ~lr; rounding of tiny gradients may flip a sign-sized step
        assert np.abs(p - z["p/" + k]).max() <= 2.5e-3, k
        assert np.mean(np.abs(p - z["p/" + k]) > 1e-5) <= 0.02, k
