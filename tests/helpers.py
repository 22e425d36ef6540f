"""Shared helpers for the parity tests (golden loader + comparisons)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUB = 61


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def check_packed(g, name, t, rtol, atol):
    """Compare tensor `t` with golden entry `name` (full, or subsample + sum + l2)."""
    a = t.detach().cpu().contiguous().numpy()
    if name in g:
        np.testing.assert_allclose(a, g[name], rtol=rtol, atol=atol, err_msg=name)
        return
    assert name + "__sub" in g, f"golden has no entry {name}"
    assert tuple(g[name + "__shape"]) == a.shape, name
    flat = a.reshape(-1)
    np.testing.assert_allclose(flat[::SUB], g[name + "__sub"], rtol=rtol, atol=atol, err_msg=name)
    l2 = float(np.sqrt((flat.astype(np.float64) ** 2).sum()))
    assert abs(l2 - float(g[name + "__l2"])) <= 1e-3 * max(float(g[name + "__l2"]), 1e-6) + atol, name


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
