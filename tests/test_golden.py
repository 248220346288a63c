"""Frozen fixtures (tests/golden/, written by tests/golden/make_golden.py).

They are outputs of the CPU oracle on seeded inputs, not of the reference
(which holds no vectors for this path and whose arithmetic is OpenCV's): they
pin the oracle against drift and give the GPU path a target that does not
depend on the oracle library being present or rebuilt."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))  # allow_pickle stays False


# ---- CPU: the oracle still reproduces its frozen outputs -----------------

def test_oracle_reproduces_knn_golden(oracle):
    g = load("knn_small.npz")
    m, c = oracle.knn_match(g["query"].astype(np.float32), g["train"].astype(np.float32), 2)
    assert (m["trainIdx"] == g["trainIdx"]).all()
    assert (m["distance"].view(np.uint32) == g["distance"].view(np.uint32)).all()
    assert (c == g["counts"]).all()
    assert (oracle.ratio_test(m, c, 0.7) == g["keep"]).all()
    # the duplicates planted in the fixture: ties resolve to ascending train index
    assert g["trainIdx"][5].tolist() == [3, 7] and g["distance"][5].tolist() == [0.0, 0.0]


@pytest.mark.parametrize("name,model", [("ransac_essential.npz", 0), ("ransac_fundamental.npz", 1)])
def test_oracle_reproduces_ransac_golden(oracle, name, model):
    g = load(name)
    K = g["K"] if model == 0 else None
    thr = float(g["threshold"])
    f = oracle.ransac(g["pts1"], g["pts2"], K, model=model, prob=0.99, threshold=thr,
                      fixed_count=int(g["hyps"]), want_counts=True)
    assert f["ok"] == int(g["fixed_ok"]) and f["n_inliers"] == int(g["fixed_n_inliers"])
    assert (f["mask"] == g["fixed_mask"]).all() and (f["counts"] == g["fixed_counts"]).all()
    assert (f["model"].view(np.uint64) == g["fixed_model"].view(np.uint64)).all()
    a = oracle.ransac(g["pts1"], g["pts2"], K, model=model, prob=0.99, threshold=thr, max_iters=300)
    assert a["iters"] == int(g["adaptive_iters"]) and (a["mask"] == g["adaptive_mask"]).all()
    assert (a["model"].view(np.uint64) == g["adaptive_model"].view(np.uint64)).all()


def test_oracle_reproduces_pose_golden(oracle):
    g = load("pose_essential.npz")
    r = oracle.recover_pose(g["E"], g["pts1"], g["pts2"], g["K"])
    assert (r["R"].view(np.uint64) == g["R"].view(np.uint64)).all()
    assert (r["t"].view(np.uint64) == g["t"].view(np.uint64)).all()
    assert (r["mask"] == g["mask"]).all() and (r["votes"] == g["votes"]).all()
    assert r["n_good"] == int(g["n_good"])
    m = oracle.recover_pose(g["E"], g["pts1"], g["pts2"], g["K"], mask=g["ransac_mask"])
    assert (m["mask"] == g["masked_mask"]).all() and (m["votes"] == g["masked_votes"]).all()
    n = oracle.recover_pose(g["E"], g["pts1"], g["pts2"], g["K"], distance_thresh=25.0)
    assert (n["mask"] == g["near_mask"]).all() and (n["votes"] == g["near_votes"]).all()


def _track_lists(g, dtype):
    ab = np.zeros(len(g["ab_query"]), dtype=dtype)
    ab["queryIdx"], ab["trainIdx"] = g["ab_query"], g["ab_train"]
    bc = np.zeros(len(g["bc_query"]), dtype=dtype)
    bc["queryIdx"], bc["trainIdx"] = g["bc_query"], g["bc_train"]
    return ab, bc


def test_oracle_reproduces_pnp_golden(oracle):
    from sfm_amd import synth
    g = load("pnp_ransac.npz")
    o = oracle.pnp_ransac(g["X"], g["u"], synth.REF_K)
    assert o["ok"] == 1 and o["iters"] == int(g["iters"]) and o["n_inliers"] == int(g["n_inliers"])
    assert (o["mask"] == g["mask"]).all()
    assert o["R"].tobytes() == g["R"].tobytes() and o["tvec"].tobytes() == g["tvec"].tobytes()
    assert o["rvec"].tobytes() == g["rvec"].tobytes()
    assert np.abs(o["R"] - g["R_true"]).max() < 2e-3 and np.abs(o["tvec"] - g["t_true"]).max() < 1.0


def test_oracle_reproduces_track_golden(oracle):
    g = load("track_join.npz")
    ab, bc = _track_lists(g, oracle.DMATCH_DTYPE)
    assert (oracle.track_join(ab, bc) == g["tracks"]).all() and len(g["tracks"]) > 100


# ---- GPU: the HIP path against the same bytes -----------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_gpu_knn_equals_golden(gpu_ctx, dtype):
    from sfm_amd import matcher
    g = load("knn_small.npz")
    m, c = matcher.BFMatcher(gpu_ctx).knnMatch(g["query"].astype(dtype), g["train"].astype(dtype), k=2)
    assert (m["trainIdx"] == g["trainIdx"]).all()
    assert (m["distance"].view(np.uint32) == g["distance"].view(np.uint32)).all()
    assert (c == g["counts"]).all()
    assert (matcher.ratio_test(m, c, 0.7) == g["keep"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,model", [("ransac_essential.npz", 0), ("ransac_fundamental.npz", 1)])
def test_gpu_ransac_equals_golden(gpu_ctx, name, model):
    from sfm_amd import matcher
    g = load(name)
    thr = float(g["threshold"])
    hyps = int(g["hyps"])
    if model == 0:
        _, mask, r = matcher.findEssentialMat(gpu_ctx, g["pts1"], g["pts2"], g["K"], prob=0.99, threshold=thr,
                                              mode=matcher.RANSAC_FIXED, n_hyp=hyps)
        _, amask, ar = matcher.findEssentialMat(gpu_ctx, g["pts1"], g["pts2"], g["K"], prob=0.99,
                                                threshold=thr, maxIters=300)
    else:
        _, mask, r = matcher.findFundamentalMat(gpu_ctx, g["pts1"], g["pts2"], ransacReprojThreshold=thr,
                                                confidence=0.99, mode=matcher.RANSAC_FIXED, n_hyp=hyps)
        _, amask, ar = matcher.findFundamentalMat(gpu_ctx, g["pts1"], g["pts2"], ransacReprojThreshold=thr,
                                                  confidence=0.99, maxIters=300)
    assert r["ok"] == int(g["fixed_ok"]) and r["n_inliers"] == int(g["fixed_n_inliers"])
    assert (mask == g["fixed_mask"]).all()
    np.testing.assert_allclose(r["model"], g["fixed_model"], rtol=1e-6, atol=0)  # stated tolerance
    assert (np.asarray(r["model"]).view(np.uint64) == g["fixed_model"].view(np.uint64)).all()  # and in fact equal
    assert ar["iters"] == int(g["adaptive_iters"]) and (amask == g["adaptive_mask"]).all()
    assert (np.asarray(ar["model"]).view(np.uint64) == g["adaptive_model"].view(np.uint64)).all()


@pytest.mark.gpu
def test_gpu_pose_equals_golden(gpu_ctx):
    from sfm_amd import matcher
    g = load("pose_essential.npz")
    good, R, t, mask = matcher.recoverPose(gpu_ctx, g["E"], g["pts1"], g["pts2"], g["K"])
    np.testing.assert_allclose(R, g["R"], rtol=0, atol=1e-6)  # stated tolerance
    assert (R.view(np.uint64) == g["R"].view(np.uint64)).all()  # and in fact equal
    assert (t.reshape(3).view(np.uint64) == g["t"].view(np.uint64)).all()
    assert good == int(g["n_good"]) and (mask == g["mask"]).all()
    assert (matcher.pose_votes(gpu_ctx) == g["votes"]).all()
    _, _, _, mm = matcher.recoverPose(gpu_ctx, g["E"], g["pts1"], g["pts2"], g["K"], mask=g["ransac_mask"])
    assert (mm == g["masked_mask"]).all() and (matcher.pose_votes(gpu_ctx) == g["masked_votes"]).all()
    _, _, _, nm = matcher.recoverPose(gpu_ctx, g["E"], g["pts1"], g["pts2"], g["K"], distanceThresh=25.0)
    assert (nm == g["near_mask"]).all() and (matcher.pose_votes(gpu_ctx) == g["near_votes"]).all()
    # the chain of main.cpp:418-420 on the GPU gives the same E, hence the same pose
    E, _, r = matcher.findEssentialMat(gpu_ctx, g["pts1"], g["pts2"], g["K"], prob=0.999, threshold=1.0)
    assert (np.asarray(E).view(np.uint64) == g["E"].view(np.uint64)).all()


@pytest.mark.gpu
def test_gpu_track_join_equals_golden(gpu_ctx):
    from sfm_amd import matcher
    g = load("track_join.npz")
    ab, bc = _track_lists(g, matcher.DMATCH_DTYPE)
    t = matcher.track_join(gpu_ctx, ab, bc)
    assert t.shape == g["tracks"].shape and (t == g["tracks"]).all()
