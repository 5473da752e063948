"""CPU model of the arithmetic the splat-parallel backward kernel (render_bwd_scan.hip) relies on.

(1) Split-bf16 operands: a per-pair scalar x goes to the matrix core as hi + lo, hi = x truncated to bf16 (top 16 bits of the
    fp32 word), lo = (x - hi) truncated to bf16; the constant operand holds dL/dpixel as hi + lo as well and the moment weights
    1, u, v, uu, uv, vv about the tile centre (half-integers below 8 and their products: exact in bf16).  With fp32 accumulation
    the products reproduce the fp32 sums to ~2^-16 relative.
(2) Moments about the tile centre -> the sums about the splat centre: with X = splat_x - tile_centre_x, u = pixel_x -
    tile_centre_x, dx = X - u:   sum g dx = X M0 - Mu,   sum g dx^2 = X^2 M0 - 2 X Mu + Muu,   sum g dx dy = XY M0 - X Mv - Y Mu + Muv.
(3) Per-pixel transmittance in front of every splat of a batch from the state BEHIND the batch and an inclusive scan of
    1 / (1 - alpha) along the batch (back to front), and R from an inclusive scan of w (c . dL/dpix)."""
import numpy as np


def bf16_trunc(x):
    x = np.asarray(x, np.float32)
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split(x):
    hi = bf16_trunc(x)
    lo = bf16_trunc(np.asarray(x, np.float32) - hi)
    return hi, lo


def test_split_bf16_products_keep_sixteen_bits():
    rng = np.random.default_rng(3)
    w = rng.uniform(1e-4, 1.0, (64, 16)).astype(np.float32) * rng.choice([1e-3, 1.0, 30.0], (64, 16)).astype(np.float32)   # [pixel, splat]
    d = rng.normal(size=(64, 3)).astype(np.float32)                                                                         # dL/dpixel
    whi, wlo = split(w)
    dhi, dlo = split(d)
    # what the four MFMAs of a half batch add up: (dhi + dlo)^T (whi + wlo) without the lo x lo term
    got = (dhi.T.astype(np.float64) @ whi + dhi.T.astype(np.float64) @ wlo + dlo.T.astype(np.float64) @ whi)
    ref = d.T.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(d).T.astype(np.float64) @ np.abs(w).astype(np.float64)      # (sums of cancelling terms: relative to the magnitudes)
    assert np.max(np.abs(got - ref) / scale) < 2.0 ** -15


def test_moment_weights_are_exact_in_bf16():
    u = np.arange(16, dtype=np.float32) - 7.5
    for m in (u, np.outer(u, u).ravel(), u * u):
        assert np.array_equal(bf16_trunc(m), m.astype(np.float32))


def test_moments_about_the_tile_centre_give_the_sums_about_the_splat():
    rng = np.random.default_rng(5)
    u, v = np.meshgrid(np.arange(16) - 7.5, np.arange(16) - 7.5)
    u, v = u.ravel(), v.ravel()
    g = rng.normal(size=256)
    X, Y = rng.uniform(-20, 20, 2)
    M0, Mu, Mv, Muu, Muv, Mvv = g.sum(), (g * u).sum(), (g * v).sum(), (g * u * u).sum(), (g * u * v).sum(), (g * v * v).sum()
    dx, dy = X - u, Y - v
    np.testing.assert_allclose(X * M0 - Mu, (g * dx).sum(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(Y * M0 - Mv, (g * dy).sum(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(X * X * M0 - 2 * X * Mu + Muu, (g * dx * dx).sum(), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(X * Y * M0 - X * Mv - Y * Mu + Muv, (g * dx * dy).sum(), rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(Y * Y * M0 - 2 * Y * Mv + Mvv, (g * dy * dy).sum(), rtol=1e-10, atol=1e-9)


def test_batch_scans_reproduce_the_sequential_recurrences():
    rng = np.random.default_rng(9)
    n = 16
    a = rng.uniform(0.0, 0.7, n) * (rng.uniform(size=n) > 0.3)      # batch in replay order (back to front), some inactive
    cd = rng.normal(size=n)                                          # c_i . dL/dpix
    T_behind, R_behind = 0.37, -0.8                                  # state left by the splats behind the batch
    # sequential replay
    T, R, Ts, Rs = T_behind, R_behind, [], []
    for i in range(n):
        T = T / (1.0 - a[i])
        Ts.append(T)
        Rs.append(R)                                                 # R behind splat i
        R = R + cd[i] * a[i] * T
    # the kernel's form: inclusive scans along the batch
    P = np.cumprod(1.0 / (1.0 - a))
    Tk = T_behind * P
    wc = cd * a * Tk
    S = np.cumsum(wc)
    Rk = (R_behind - wc) + S                                         # exclusive: the batch's earlier splats + the state
    np.testing.assert_allclose(Tk, Ts, rtol=1e-12)
    np.testing.assert_allclose(Rk, Rs, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(T_behind * P[-1], T, rtol=1e-12)      # lane 15 stores the state for the next batch
    np.testing.assert_allclose(R_behind + S[-1], R, rtol=1e-10)
