"""GPU parity tests for distCUDA2 (simple_knn._C): bit-exact against the exhaustive fp32 oracle, and against
scipy.spatial.cKDTree at sizes the exhaustive scan cannot reach."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(pts):
    from simple_knn._C import distCUDA2
    out = distCUDA2(torch.from_numpy(pts).cuda())
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("kind,P", [("uniform", 5000), ("uniform", 20000), ("grid", 64 * 48), ("clustered", 12000), ("tiny", 4),
                                    ("tiny", 7), ("plane", 9000)])
def test_knn_bit_exact_vs_oracle(kind, P):
    from oracle import c_oracle
    g = np.random.default_rng(100 + P)
    if kind == "uniform":
        pts = g.random((P, 3), dtype=np.float32) * np.array([4.0, 2.0, 9.0], np.float32) - 1.0
    elif kind == "grid":            # the DAS3R case: un-projected pixel grid (exact ties in distance)
        ys, xs = np.meshgrid(np.arange(48, dtype=np.float32), np.arange(64, dtype=np.float32), indexing="ij")
        z = 2.0 + 0.01 * g.random((48, 64), dtype=np.float32)
        pts = np.stack([(xs - 32) * z / 50.0, (ys - 24) * z / 50.0, z], -1).reshape(-1, 3).astype(np.float32)
    elif kind == "clustered":
        c = g.random((30, 3), dtype=np.float32) * 10
        pts = (c[g.integers(0, 30, P)] + 0.05 * g.standard_normal((P, 3)).astype(np.float32)).astype(np.float32)
        pts[500:520] = pts[0]       # exact duplicates: distance 0 neighbours
    elif kind == "plane":
        pts = g.random((P, 3), dtype=np.float32)
        pts[:, 2] = 1.5
    else:
        pts = g.random((P, 3), dtype=np.float32)
    got = _run(pts)
    ref = c_oracle.knn3_mean_dist2(pts)
    assert np.array_equal(got, ref), f"max abs diff {np.abs(got - ref).max()}"


def test_knn_large_vs_kdtree():
    from scipy.spatial import cKDTree
    P = 300_000
    g = np.random.default_rng(7)
    pts = g.random((P, 3), dtype=np.float32) * np.array([3.0, 2.0, 8.0], np.float32)
    got = _run(pts)
    p64 = pts.astype(np.float64)
    d, _ = cKDTree(p64).query(p64, k=4, workers=-1)
    ref = (d[:, 1:] ** 2).mean(1)
    np.testing.assert_allclose(got, ref, rtol=3e-5, atol=1e-12)


def test_knn_is_permutation_equivariant():
    g = np.random.default_rng(11)
    pts = g.random((8000, 3), dtype=np.float32)
    perm = g.permutation(8000)
    a = _run(pts)
    b = _run(np.ascontiguousarray(pts[perm]))
    assert np.array_equal(a[perm], b)


def test_das3r_scale_init_recipe():
    """scene/gaussian_model.py:641-642: dist2 = clamp_min(distCUDA2(pts), 1e-7); scales = log(sqrt(dist2))."""
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(4096, 3, generator=g).cuda()
    dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
    assert scales.shape == (4096, 3) and torch.isfinite(scales).all()


def das3r_shaped_points(frames=20, H=208, W=512, seed=3):
    """The point set DAS3R calls distCUDA2 on (scene/gaussian_model.py:641): every pixel of every frame un-projected with its
    depth, frames seen from slightly different poses -> 2.1 M points on `frames` nearly coincident depth sheets."""
    g = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    out = []
    for f in range(frames):
        z = (4.0 + np.sin(xs / 37.0 + f) * 0.6 + np.cos(ys / 23.0) * 0.4 + 0.02 * g.random((H, W), dtype=np.float32)).astype(np.float32)
        p = np.stack([(xs - W / 2) * z / 600.0 + 0.05 * f, (ys - H / 2) * z / 600.0, z], -1)
        out.append(p.reshape(-1, 3))
    return np.concatenate(out).astype(np.float32)


def test_knn_at_das3r_scale_vs_kdtree():
    """2.1 M points in the DAS3R shape (the largest tested P was 300 k) against scipy's exact k-d tree."""
    from scipy.spatial import cKDTree
    pts = das3r_shaped_points()
    assert pts.shape[0] == 20 * 208 * 512
    got = _run(pts)
    p64 = pts.astype(np.float64)
    d, _ = cKDTree(p64).query(p64, k=4, workers=-1)
    ref = (d[:, 1:] ** 2).mean(1)
    np.testing.assert_allclose(got, ref, rtol=5e-5, atol=1e-12)
