"""CPU statement of the f16x2 operand format (csrc/common.h::split2_f16, csrc/gemm_bf3.hip): what the two fp16 planes represent
and what the three-product contraction loses, in numpy (IEEE round-to-nearest-even conversions) - no GPU needed.  The GPU suite
pins the device kernels to exactly this arithmetic (tests/test_gemm_gpu.py::test_split_f16x2_planes_equal_the_numpy_statement)."""
import numpy as np


def split2_f16(x: np.ndarray, scale: float):
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    h1 = xs.astype(np.float16)
    h2 = (xs - h1.astype(np.float32)).astype(np.float16)
    return h1, h2


def test_two_planes_represent_the_scaled_value_to_2_pow_minus_22():
    rng = np.random.default_rng(0)
    # magnitudes from the top of the range (after scale 4: up to 16376) down to where h2 goes subnormal and below
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-18, 8.3, 200000))).astype(np.float32)
    x = x[np.abs(x) < 16376]
    h1, h2 = split2_f16(x, 4.0)
    assert np.isfinite(h1.astype(np.float32)).all() and np.isfinite(h2.astype(np.float32)).all()
    xs = x.astype(np.float64) * 4.0
    err = np.abs(xs - h1.astype(np.float64) - h2.astype(np.float64))
    # relative 2^-22 wherever h2 is a normal fp16 number, and never more than half a subnormal step (2^-25) in absolute terms below that
    bound = np.maximum(np.abs(xs) * 2.0 ** -22, 2.0 ** -25)
    assert (err <= bound).all(), float((err / bound).max())
    # ... and the first plane alone is the fp16 rounding of the value (11 significand bits)
    big = np.abs(xs) >= 2.0 ** -14
    assert (np.abs(xs - h1.astype(np.float64))[big] <= np.abs(xs)[big] * 2.0 ** -11).all()


def test_out_of_range_values_become_inf_not_a_clamped_number():
    h1, _ = split2_f16(np.array([16376.0, 16384.0, -2.0e4], np.float32), 4.0)
    assert np.isfinite(h1[0]) and np.isinf(h1[1]) and np.isinf(h1[2])


def test_three_products_lose_only_the_product_of_the_second_planes():
    rng = np.random.default_rng(1)
    K = 2304
    a = np.maximum(rng.standard_normal((64, K)) * 1.5 + 0.3, 0).astype(np.float32)          # post-ReLU-like activations
    w = (rng.standard_normal((32, K)) / np.sqrt(K)).astype(np.float32)
    sw = 2.0 ** np.floor(14 - np.log2(np.abs(w).max()))
    a1, a2 = [p.astype(np.float64) for p in split2_f16(a, 4.0)]
    w1, w2 = [p.astype(np.float64) for p in split2_f16(w, sw)]
    three = (a1 @ w1.T + a1 @ w2.T + a2 @ w1.T) / (4.0 * sw)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(exact).max()
    # in exact arithmetic the three products are within a few 2^-22 of the true product (representation of both operands + the
    # dropped h2*h2'): far below what fp32 accumulation of K = 2304 terms adds on the GPU (~1e-6 of scale)
    assert np.abs(three - exact).max() <= 3e-7 * scale, np.abs(three - exact).max() / scale
    dropped = np.abs((a2 @ w2.T) / (4.0 * sw)).max()
    assert dropped <= 2.0 ** -22 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T).max()
