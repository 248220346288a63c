"""Resection (cv::solvePnPRansac with SOLVEPNP_EPNP, bundle_adjustment.h:139-155)
on the GPU against the CPU oracle, through the C ABI.

Bar: same verdict, same iteration count, the same inlier mask byte for byte, and
R, t, rvec equal to the oracle's in every bit (both run the same unfused IEEE
double operations in the same order).  The oracle itself is pinned only by known
answers (planted poses); OpenCV's own EPnP is not available: parity unpinned.
"""
import numpy as np
import pytest

from sfm_amd import _lib, matcher, synth

pytestmark = pytest.mark.gpu
K = synth.REF_K


def scene(n, noise, out_frac, seed, planar=False):
    rng = np.random.default_rng(seed)
    R = synth._rot(rng.normal(size=3), np.deg2rad(rng.uniform(5, 40)))
    t = np.array([rng.normal() * 30, rng.normal() * 30, 600 + rng.normal() * 50])
    X = rng.uniform(-150, 150, (n, 3))
    if planar:
        X[:, 2] = 0.02 * X[:, 0]
    Xc = X @ R.T + t
    u = Xc @ K.T
    u = u[:, :2] / u[:, 2:]
    u += rng.normal(0, noise, u.shape)
    no = int(n * out_frac)
    u[:no] = rng.uniform(0, 1500, (no, 2))
    perm = rng.permutation(n)
    return X[perm], u[perm], R, t


def _same(gpu, o):
    ok, rvec, tvec, inl, info = gpu
    assert int(ok) == o["ok"]
    if not o["ok"]:
        return
    assert info["iters"] == o["iters"] and info["n_inliers"] == o["n_inliers"]
    assert (info["mask"] == o["mask"]).all()
    assert inl.tolist() == np.flatnonzero(o["mask"]).tolist()
    assert info["R"].tobytes() == o["R"].tobytes()
    assert tvec.tobytes() == o["tvec"].tobytes() and rvec.tobytes() == o["rvec"].tobytes()


@pytest.mark.parametrize("n,noise,out,seed", [(6, 0.0, 0.0, 1), (40, 0.5, 0.2, 2), (700, 1.0, 0.5, 3),
                                              (3000, 0.5, 0.3, 4), (12000, 0.3, 0.1, 5), (257, 2.0, 0.7, 6)])
def test_pnp_ransac_matches_oracle(gpu_ctx, oracle, n, noise, out, seed):
    X, u, R, t = scene(n, noise, out, seed)
    o = oracle.pnp_ransac(X, u, K)
    g = matcher.solvePnPRansac(gpu_ctx, X, u, K)
    _same(g, o)
    if out <= 0.5:
        assert g[0] and np.abs(g[4]["R"] - R).max() < 5e-3 and np.abs(g[2] - t).max() < 2.0
        assert g[4]["n_inliers"] >= 0.9 * (n - int(n * out))
        # the rotation vector is the one cv::Rodrigues inverts
        th = np.linalg.norm(g[1])
        k = g[1] / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        assert np.abs(np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx - g[4]["R"]).max() < 1e-9


@pytest.mark.parametrize("iters,err,conf", [(1, 8.0, 0.99), (7, 2.0, 0.5), (100, 0.5, 0.999999), (1000, 20.0, 0.9)])
def test_pnp_ransac_parameters(gpu_ctx, oracle, iters, err, conf):
    X, u, _, _ = scene(1500, 0.7, 0.4, 11)
    o = oracle.pnp_ransac(X, u, K, iterations_count=iters, reprojection_error=err, confidence=conf)
    _same(matcher.solvePnPRansac(gpu_ctx, X, u, K, iters, err, conf), o)


def test_pnp_degenerate_inputs(gpu_ctx, oracle):
    # exactly five points: no RANSAC, one direct solve, every point an inlier
    X, u, R, t = scene(5, 0.0, 0.0, 21)
    o = oracle.pnp_ransac(X, u, K)
    g = matcher.solvePnPRansac(gpu_ctx, X, u, K)
    _same(g, o)
    assert g[0] and g[4]["mask"].tolist() == [1] * 5 and np.abs(g[4]["R"] - R).max() < 1e-6   # float32 narrowing of the inputs
    # planar object, repeated points, all outliers: same verdict as the oracle, no fault
    for Xd, ud in (scene(400, 0.5, 0.2, 22, planar=True)[:2],
                   (np.tile(X[:1], (60, 1)), np.tile(u[:1], (60, 1))),
                   scene(300, 0.5, 1.0, 23)[:2]):
        _same(matcher.solvePnPRansac(gpu_ctx, Xd, ud, K), oracle.pnp_ransac(Xd, ud, K))
    for nbad, code in ((3, _lib.ERR_ARG), (4, _lib.ERR_UNSUPPORTED)):
        with pytest.raises(matcher.SfmGpuError) as e:
            matcher.solvePnPRansac(gpu_ctx, X[:nbad], u[:nbad], K)
        assert e.value.code == code
    with pytest.raises(matcher.SfmGpuError):
        matcher.solvePnPRansac(gpu_ctx, X, u, K, reprojectionError=0.0)


def test_pnp_golden(gpu_ctx):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pnp_ransac.npz"))
    ok, rvec, tvec, inl, info = matcher.solvePnPRansac(gpu_ctx, g["X"], g["u"], K)
    assert ok and info["iters"] == int(g["iters"]) and (info["mask"] == g["mask"]).all()
    assert info["R"].tobytes() == g["R"].tobytes() and tvec.tobytes() == g["tvec"].tobytes()
    assert rvec.tobytes() == g["rvec"].tobytes()
