"""GPU pose recovery (sfm_gpu_recover_pose, the recoverPose call of
main.cpp:420 / bundle_adjustment.h:80) against the CPU oracle: rotation,
translation, mask and all four vote counts must be identical."""
import ctypes

import numpy as np
import pytest

from sfm_amd import synth

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return (np.ascontiguousarray(a, dtype=np.float64).reshape(-1).view(np.uint64) ==
            np.ascontiguousarray(b, dtype=np.float64).reshape(-1).view(np.uint64)).all()


def check(gpu_ctx, oracle, E, p1, p2, K, dist=50.0, mask=None):
    from sfm_amd import matcher
    good, R, t, m = matcher.recoverPose(gpu_ctx, E, p1, p2, K, distanceThresh=dist, mask=mask)
    o = oracle.recover_pose(E, p1, p2, K, distance_thresh=dist, mask=mask)
    assert good == o["n_good"]
    assert same_bits(R, o["R"]) and same_bits(t, o["t"])
    assert (m == o["mask"]).all()
    assert (matcher.pose_votes(gpu_ctx) == o["votes"]).all()
    return good, R, t, m


@pytest.mark.parametrize("n", [0, 1, 63, 64, 255, 256, 257, 1000, 8192, 100003])
def test_pose_sizes(gpu_ctx, oracle, n):
    p1, p2, is_in, E, R, t = synth.two_view(n, 100 + n % 97, inlier_frac=0.85, with_pose=True)
    good, Rg, tg, m = check(gpu_ctx, oracle, E, p1, p2, synth.REF_K)
    if n >= 63:
        assert np.abs(Rg - R).max() < 1e-9 and np.abs(tg.reshape(3) - t).max() < 1e-9
        assert m[is_in].mean() > 0.99


@pytest.mark.parametrize("seed", range(6))
def test_pose_after_gpu_ransac(gpu_ctx, oracle, seed):
    """main.cpp:418-420: findEssentialMat(ptsL, ptsR, K, RANSAC) then recoverPose."""
    from sfm_amd import matcher
    p1, p2, is_in, Etrue, R, t = synth.two_view(900, 40 + seed, inlier_frac=0.7, noise_px=0.4, with_pose=True)
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.999, threshold=1.0)
    assert r["ok"] == 1
    good, Rg, tg, m = check(gpu_ctx, oracle, E, p1, p2, synth.REF_K)
    # the estimated pose is the planted one up to the noise
    ang = np.degrees(np.arccos(np.clip((np.trace(Rg @ R.T) - 1) / 2, -1, 1)))
    assert ang < 0.5 and float(tg.reshape(3) @ t) > 0.99
    # restricted to the RANSAC inliers
    check(gpu_ctx, oracle, E, p1, p2, synth.REF_K, mask=mask)


def test_pose_sign_swap_and_threshold(gpu_ctx, oracle):
    p1, p2, is_in, E, R, t = synth.two_view(3000, 9, inlier_frac=0.8, with_pose=True)
    K = synth.REF_K
    _, Ra, ta, ma = check(gpu_ctx, oracle, E, p1, p2, K)
    _, Rb, tb, mb = check(gpu_ctx, oracle, -E, p1, p2, K)
    assert np.abs(Ra - Rb).max() < 1e-12 and np.abs(ta - tb).max() < 1e-12 and (ma == mb).all()
    _, Rc, tc, _ = check(gpu_ctx, oracle, E.T, p2, p1, K)
    assert np.abs(Rc - R.T).max() < 1e-9
    for dist in (10.0, 20.0, 27.0, 36.0, 1e9):
        check(gpu_ctx, oracle, E, p1, p2, K, dist=dist)
    # scaled E (findEssentialMat's output has unit Frobenius norm; callers may rescale)
    check(gpu_ctx, oracle, 37.5 * E, p1, p2, K)
    # a second camera matrix
    K2 = np.array([[1500.0, 0, 640], [0, 1480.0, 360], [0, 0, 1]])
    check(gpu_ctx, oracle, E, p1, p2, K2)


def test_pose_degenerate_points(gpu_ctx, oracle):
    """Identical rays (a point at infinity), points on the epipole and exact
    zeros take the same branch on both sides."""
    _, _, _, E, R, t = synth.two_view(10, 1, with_pose=True)
    K = np.eye(3)
    rng = np.random.default_rng(2)
    p1 = rng.uniform(-1, 1, (512, 2))
    p2 = p1.copy()                      # pure-rotation-like correspondences
    p2[::2] = rng.uniform(-1, 1, (256, 2))
    p1[5] = p2[5] = 0.0
    check(gpu_ctx, oracle, E, p1, p2, K)
    ident = np.array([[0.0, 0, 0], [0, 0, -1], [0, 1, 0]])  # [t]x with t = x axis, R = I
    check(gpu_ctx, oracle, ident, p1, p1, K)                # identical rays under the identity candidate


def test_pose_errors(gpu_ctx):
    from sfm_amd import matcher
    p1, p2, _, E = synth.two_view(20, 1)
    with pytest.raises(matcher.SfmGpuError) as ei:
        matcher.recoverPose(gpu_ctx, np.zeros((3, 3)), p1, p2, synth.REF_K)
    assert "rank" in str(ei.value)
    with pytest.raises(matcher.SfmGpuError):
        matcher.recoverPose(gpu_ctx, np.outer([1.0, 2, 3], [3.0, 2, 1]), p1, p2, synth.REF_K)
    with pytest.raises(matcher.SfmGpuError):
        matcher.recoverPose(gpu_ctx, E, p1, p2, np.zeros((3, 3)))
    L = gpu_ctx._L
    assert L.sfm_gpu_recover_pose(gpu_ctx.handle, None, None, None, 0, None, 50.0, None, None, None, None) < 0
    # the context still works afterwards
    good, R, t, m = matcher.recoverPose(gpu_ctx, E, p1, p2, synth.REF_K)
    assert good > 0
