"""CPU model of the bucket-parallel backward replay (render_bwd_scan.hip + the forward's checkpoints, common.h BUCKET).

One pixel, a list of n splats with alphas a_i and colours c_i (3 channels), background bg, upstream gradient dLp (3).  The
sequential backward of SURVEY.md A.7 walks the list back to front with two scalars, T (transmittance in front of the current
splat) and R (sum over the splats BEHIND it of (c_j . dLp) a_j T_j):
    dL/da_i = T_i (c_i . dLp) - (R_i + T_final (bg . dLp)) / (1 - a_i)
The bucket-parallel kernel starts bucket b (list positions [b B, (b + 1) B)) from the forward's checkpoint in front of position
(b + 1) B:   T = ckpt.T,   R = dLp . (C_final - C_ckpt)   (C = colour accumulated so far, without the background)
and replays only its own positions.  This test holds that formulation to the sequential one in float64, for lists that end
inside a bucket, on a bucket boundary, and for a single bucket."""
import numpy as np
import pytest


def forward(a, c, B):
    """-> (T_final, C_final, checkpoints): checkpoint s = (T, C) in front of position (s + 1) B, last = final values."""
    T, C, ck = 1.0, np.zeros(3), []
    for i in range(len(a)):
        if i > 0 and i % B == 0:
            ck.append((T, C.copy()))
        C = C + c[i] * a[i] * T
        T = T * (1.0 - a[i])
    ck.append((T, C.copy()))
    return T, C, ck


def backward_sequential(a, c, dLp, bg, T_final):
    n = len(a)
    g = np.zeros(n)
    T, R = T_final, 0.0
    for i in range(n - 1, -1, -1):
        T = T / (1.0 - a[i])                       # transmittance in front of splat i
        cd = float(c[i] @ dLp)
        g[i] = T * cd - (R + T_final * float(bg @ dLp)) / (1.0 - a[i])
        R = R + cd * a[i] * T
    return g


def backward_buckets(a, c, dLp, bg, T_final, C_final, ck, B):
    n = len(a)
    g = np.zeros(n)
    nb = (n + B - 1) // B
    for b in range(nb):                            # every bucket on its own, in any order
        lo, hi = b * B, min(n, (b + 1) * B)
        if b < nb - 1:
            Tc, Cc = ck[b]
            T, R = Tc, float(dLp @ (C_final - Cc))
        else:
            T, R = T_final, 0.0
        for i in range(hi - 1, lo - 1, -1):
            T = T / (1.0 - a[i])
            cd = float(c[i] @ dLp)
            g[i] = T * cd - (R + T_final * float(bg @ dLp)) / (1.0 - a[i])
            R = R + cd * a[i] * T
    return g


@pytest.mark.parametrize("n,B", [(1, 8), (8, 8), (9, 8), (37, 8), (64, 8), (100, 16), (5, 1024)])
def test_bucket_replay_equals_sequential_replay(n, B):
    rng = np.random.default_rng(n * 131 + B)
    a = rng.uniform(0.004, 0.6, n)
    c = rng.uniform(0.0, 1.0, (n, 3))
    dLp, bg = rng.normal(size=3), rng.uniform(0, 1, 3)
    T_final, C_final, ck = forward(a, c, B)
    assert len(ck) == (n + B - 1) // B
    ref = backward_sequential(a, c, dLp, bg, T_final)
    got = backward_buckets(a, c, dLp, bg, T_final, C_final, ck, B)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12)


def test_checkpoint_is_the_forward_state_in_front_of_the_boundary():
    rng = np.random.default_rng(7)
    a, c = rng.uniform(0.01, 0.5, 40), rng.uniform(0, 1, (40, 3))
    _, _, ck = forward(a, c, 16)
    T = np.prod(1.0 - a[:16])
    np.testing.assert_allclose(ck[0][0], T, rtol=1e-12)
    np.testing.assert_allclose(ck[1][0], np.prod(1.0 - a[:32]), rtol=1e-12)
