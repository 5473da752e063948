"""CPU model of the arithmetic the block-list backward kernel (das3r_amd/csrc/render_bwd_blk.hip) relies on.

(1) Moments of g about the block's CORNER pixel (weights u, v in 0..3) -> the sums about the splat centre, with X = splat_x -
    corner_x, dx = X - u.
(2) position < n_contrib folded into the alpha clamp: with integers lastrel in [0, MB], posrel in [0, MB - 1],
    t = lastrel - posrel is >= 1 exactly where posrel < lastrel and <= 0 elsewhere, so min(alpha, t) >= 1/255 <=> both tests pass
    (alpha <= 0.99 < 1), and where it passes min(alpha, t) == alpha.
(3) The octagon test (axis-aligned extents + the two diagonal ones, the diagonal ones derived from the axis-aligned ones and the
    conic without the determinant) never rejects a block the ellipse reaches.
(4) The per-row scans + the hand-off of the row totals to the pixel lane reproduce the sequential back-to-front replay over a
    block's list split into batches of 16 (as tests/test_scan_model.py does for one batch)."""
import numpy as np


def test_corner_moments_give_the_sums_about_the_splat():
    rng = np.random.default_rng(11)
    u, v = np.meshgrid(np.arange(4.0), np.arange(4.0))
    u, v = u.ravel(), v.ravel()
    for _ in range(20):
        g = rng.normal(size=16)
        X, Y = rng.uniform(-40, 40, 2)
        M0, Mu, Mv, Muu, Muv, Mvv = g.sum(), (g * u).sum(), (g * v).sum(), (g * u * u).sum(), (g * u * v).sum(), (g * v * v).sum()
        dx, dy = X - u, Y - v
        np.testing.assert_allclose(X * M0 - Mu, (g * dx).sum(), rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(Y * M0 - Mv, (g * dy).sum(), rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(X * X * M0 - 2 * X * Mu + Muu, (g * dx * dx).sum(), rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(X * Y * M0 - X * Mv - Y * Mu + Muv, (g * dx * dy).sum(), rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(Y * Y * M0 - 2 * Y * Mv + Mvv, (g * dy * dy).sum(), rtol=1e-10, atol=1e-8)


def test_last_contributor_folds_into_the_alpha_clamp():
    MB = 128
    rng = np.random.default_rng(12)
    for _ in range(2000):
        lo, max_contrib, done_before = int(rng.integers(0, 5000)), int(rng.integers(1, 3000)), 0
        done_before = MB * int(rng.integers(0, (max_contrib + MB - 1) // MB))
        n_contrib = int(rng.integers(0, lo + max_contrib + 300))
        base = lo + max_contrib - done_before - MB
        lastrel = float(min(max(n_contrib - base, 0), MB))
        j = int(rng.integers(0, min(MB, max_contrib - done_before)))
        position = lo + max_contrib - 1 - done_before - j
        posrel = float(MB - 1 - j)
        alpha = np.float32(rng.choice([0.0, 1.0 / 255.0, 0.0039, 0.3, 0.99]))
        t = np.float32(lastrel - posrel)
        a1 = min(alpha, t)
        want = (position < n_contrib) and (alpha >= np.float32(1.0 / 255.0))
        assert (a1 >= np.float32(1.0 / 255.0)) == want
        if want:
            assert a1 == alpha


def _min_q_rect(A, B, C, dxl, dxh, dyl, dyh):
    def qf(dx, dy):
        return A * dx * dx + 2 * B * dx * dy + C * dy * dy
    if dxl <= 0 <= dxh and dyl <= 0 <= dyh:
        return 0.0
    return min(qf(dxl, np.clip(-B * dxl / C, dyl, dyh)), qf(dxh, np.clip(-B * dxh / C, dyl, dyh)),
               qf(np.clip(-B * dyl / A, dxl, dxh), dyl), qf(np.clip(-B * dyh / A, dxl, dxh), dyh))


def test_octagon_test_is_conservative():
    rng = np.random.default_rng(13)
    rejected = 0
    for _ in range(20000):
        # random covariance: sigmas 0.3 .. 300 px, any rotation
        s1, s2 = np.exp(rng.uniform(np.log(0.3), np.log(300.0), 2))
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        Sig = R @ np.diag([s1 * s1, s2 * s2]) @ R.T + 0.3 * np.eye(2)
        con = np.linalg.inv(Sig)
        A, B, C = np.float32(con[0, 0]), np.float32(con[0, 1]), np.float32(con[1, 1])
        o = np.float32(rng.uniform(0.005, 1.0))
        if 255.0 * o <= 1.0:
            continue
        tau2 = np.float32(2.0 * np.log(255.0 * o) * 1.0005 + 1e-3)
        hx = np.float32(np.sqrt(tau2 * np.float32(Sig[0, 0])) * 1.0005 + 0.02)       # preprocess.hip
        hy = np.float32(np.sqrt(tau2 * np.float32(Sig[1, 1])) * 1.0005 + 0.02)
        # the kernel's fp32 derivation of the diagonal extents
        ex, ey = np.float32((hx - np.float32(0.02)) / np.float32(1.0005)), np.float32((hy - np.float32(0.02)) / np.float32(1.0005))
        ex2, ey2 = np.float32(ex * ex), np.float32(ey * ey)
        txy = np.float32(-B * ex2 / C)
        half, slack = np.float32(0.5) * (ex2 + ey2), np.float32(2e-6) * (ex2 + ey2) + np.float32(1e-3)
        hd1 = np.float32(np.sqrt(max(half + txy, np.float32(0)) + slack) * 1.0005 + 0.05)
        hd2 = np.float32(np.sqrt(max(half - txy, np.float32(0)) + slack) * 1.0005 + 0.05)
        # a block near the edge of the footprint
        reach = 3.5 * max(s1, s2)
        x, y = rng.uniform(-reach, reach, 2)
        cx, cy = 1.5, 1.5                                     # block of pixel centres [0, 3] x [0, 3]
        ddx, ddy = x - cx, y - cy
        hit = (abs(ddx) <= hx + 1.5 and abs(ddy) <= hy + 1.5 and abs(ddx + ddy) * 0.70710678 <= hd1 + 2.1213204 + 1e-3
               and abs(ddx - ddy) * 0.70710678 <= hd2 + 2.1213204 + 1e-3)
        # exact: can alpha reach 1/255 on the block's rectangle?  (float64, true tau without margins)
        qmin = _min_q_rect(float(con[0, 0]), float(con[0, 1]), float(con[1, 1]), x - 3, x - 0, y - 3, y - 0)
        reaches = qmin <= 2.0 * np.log(255.0 * float(o))
        if reaches:
            assert hit, (s1, s2, th, o, x, y)
        rejected += (not hit)
    assert rejected > 1000, "the test should reject something"


def test_block_walk_reproduces_the_sequential_replay():
    rng = np.random.default_rng(14)
    npx, nsplat = 16, 53                                       # one block, a list of 53 entries in replay order (back to front)
    alpha = rng.uniform(0.0, 0.7, (npx, nsplat)) * (rng.uniform(size=(npx, nsplat)) > 0.4)
    cd = rng.normal(size=(npx, nsplat))
    G = rng.uniform(0.1, 1.0, (npx, nsplat))
    tfbg = rng.normal(size=npx) * 0.1
    T_final = rng.uniform(0.05, 0.9, npx)
    # sequential (upstream order): g[p, i] = G * (T_i cd - (R + tfbg) / (1 - alpha))
    g_ref = np.zeros((npx, nsplat))
    w_ref = np.zeros((npx, nsplat))
    for p in range(npx):
        T, R = T_final[p], 0.0
        for i in range(nsplat):
            a = alpha[p, i]
            T = T / (1.0 - a)
            w = a * T
            g_ref[p, i] = (G[p, i] if a > 0 else 0.0) * (T * cd[p, i] - (R + tfbg[p]) / (1.0 - a))
            w_ref[p, i] = w
            R += cd[p, i] * w
    # the kernel's form: batches of 16 lanes; per pixel step inclusive scans along the lanes, totals back to the pixel lane
    stT, stR = T_final.copy(), np.zeros(npx)
    g_k = np.zeros((npx, nsplat))
    w_k = np.zeros((npx, nsplat))
    for b in range(0, nsplat, 16):
        lanes = np.arange(b, min(b + 16, nsplat))
        pad = 16 - len(lanes)
        for p in range(npx):                                   # the 16 pixel steps
            a = np.concatenate([alpha[p, lanes], np.zeros(pad)])
            c = np.concatenate([cd[p, lanes], np.zeros(pad)])
            Gm = np.concatenate([np.where(alpha[p, lanes] > 0, G[p, lanes], 0.0), np.zeros(pad)])
            rinv = 1.0 / (1.0 - a)
            T = stT[p] * np.cumprod(rinv)
            w = a * T
            wc = c * w
            Rinc = stR[p] + np.cumsum(wc)
            Rex = Rinc - wc
            g = Gm * (T * c - (Rex + tfbg[p]) * rinv)
            g_k[p, lanes] = g[:len(lanes)]
            w_k[p, lanes] = w[:len(lanes)]
            stT[p], stR[p] = T[15], Rinc[15]
    np.testing.assert_allclose(g_k, g_ref, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(w_k, w_ref, rtol=1e-10, atol=1e-14)
