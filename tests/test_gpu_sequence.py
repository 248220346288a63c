"""What the reference does with the output of match() for a whole sequence
(main.cpp:400-456), batched: the pose of every pair from its own essential
matrix (sfm_gpu_match_pairs_pose) and the three-view joins of consecutive pairs
(sfm_gpu_track_join_sequence), against the CPU oracle called pair by pair."""
import ctypes

import numpy as np
import pytest

from sfm_amd import _lib, matcher, synth

pytestmark = pytest.mark.gpu


def _ring(n_img, n_desc, n_planted, seed):
    """Images in a ring, image v+1 holding noisy copies of rows of image v at
    keypoints that follow one two-view geometry per pair."""
    rng = np.random.default_rng(seed)
    descs = [synth.sift_like(n_desc, seed + 1 + v) for v in range(n_img)]
    kps = [np.stack([rng.uniform(0, synth.IMG_W, n_desc), rng.uniform(0, synth.IMG_H, n_desc)], 1).astype(np.float32)
           for _ in range(n_img)]
    free = [np.ones(n_desc, bool) for _ in range(n_img)]
    for v in range(n_img - 1):
        w = v + 1
        rows_q = rng.permutation(np.flatnonzero(free[v]))[:n_planted]
        rows_t = rng.permutation(np.flatnonzero(free[w]))[:n_planted]
        descs[w][rows_t] = np.clip(descs[v][rows_q] + np.rint(rng.normal(0, 6, (n_planted, 128))), 0, 255)
        planted = np.full(n_desc, -1, np.int64)
        planted[rows_q] = rows_t
        kq, kt = synth.keypoints_for_pair(n_desc, n_desc, planted, seed + 100 + v)
        kps[v][rows_q] = kq[rows_q]
        kps[w][rows_t] = kt[rows_t]
        free[v][rows_q] = False   # these rows' keypoints are fixed now
        free[w][rows_t] = False
    return [d.astype(np.float32) for d in descs], kps


def _oracle_pair(oracle, dq, dt, kq, kt, K, **kw):
    m, c = oracle.knn_match(dq, dt, 2, nthreads=8)
    good = m[oracle.ratio_test(m, c, 0.7), 0]
    p1 = kq[good["queryIdx"]].astype(np.float64)
    p2 = kt[good["trainIdx"]].astype(np.float64)
    r = oracle.ransac(p1, p2, K, model=0, **kw)
    inl = r["mask"].astype(bool) if r["ok"] else np.zeros(len(good), bool)
    return r, good[inl], p1[inl], p2[inl]


@pytest.mark.parametrize("mode", ["adaptive", "fixed"])
def test_match_pairs_pose_equals_recover_pose_per_pair(gpu_ctx, oracle, mode):
    n_img, n_desc = 4, 2200
    descs, kps = _ring(n_img, n_desc, 700, 4100)
    # a last image nothing was planted into: its pair has no model
    descs.append(synth.sift_like(600, 4999).astype(np.float32))
    kps.append(np.random.default_rng(3).uniform(0, 3000, (600, 2)).astype(np.float32))
    imgs = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    pairs = np.array([[0, 1], [1, 2], [2, 3], [3, 4]], np.int32)
    kw = dict(prob=0.99, threshold=1.0)
    if mode == "fixed":
        res = matcher.match_pairs_pose(gpu_ctx, imgs, pairs, synth.REF_K, mode=matcher.RANSAC_FIXED, n_hyp=160,
                                       want_matches=True, **kw)
        okw = dict(fixed_count=160, **kw)
    else:
        res = matcher.match_pairs_pose(gpu_ctx, imgs, pairs, synth.REF_K, want_matches=True, **kw)
        okw = dict(max_iters=1000, **kw)
    with_pose = 0
    for p, (i, j) in enumerate(pairs):
        r, sel, p1, p2 = _oracle_pair(oracle, descs[i], descs[j], kps[i], kps[j], synth.REF_K, **okw)
        assert res["n_inliers"][p] == len(sel)
        assert (res["inlier_matches"][p]["trainIdx"] == sel["trainIdx"]).all()
        if len(sel) == 0:
            assert res["n_pose_good"][p] == -1 and not res["R"][p].any() and not res["t"][p].any()
            continue
        with_pose += 1
        # main.cpp:418-420: a second RANSAC (OpenCV's defaults: 0.999, 1 px, 1000 iterations)
        # over the selected matches gives the matrix recoverPose decomposes, and every
        # selected match votes
        r2 = oracle.ransac(p1, p2, synth.REF_K, model=0, prob=0.999, threshold=1.0, max_iters=1000)
        assert r2["ok"] == 1
        assert res["E"][p].tobytes() == np.asarray(r2["model"]).tobytes()
        o = oracle.recover_pose(r2["model"], p1, p2, synth.REF_K)
        assert res["n_pose_good"][p] == o["n_good"]
        assert res["R"][p].tobytes() == o["R"].tobytes()
        assert res["t"][p].tobytes() == o["t"].reshape(3).tobytes()
        # and the library's own one-pair call on the same inliers
        good, R, t, _ = matcher.recoverPose(gpu_ctx, res["E"][p], p1, p2, synth.REF_K)
        assert good == res["n_pose_good"][p] and R.tobytes() == res["R"][p].tobytes()
        assert o["n_good"] > 0.9 * len(sel)
    assert with_pose == 3 and res["n_pose_good"][3] == -1
    for im in imgs:
        im.close()


def test_match_pairs_pose_argument_checks(gpu_ctx):
    d = synth.sift_like(300, 1).astype(np.float32)
    k = np.random.default_rng(1).uniform(0, 1000, (300, 2)).astype(np.float32)
    imgs = [matcher.Image(gpu_ctx, d, k), matcher.Image(gpu_ctx, d, k)]
    L = gpu_ctx._L
    arr = (ctypes.c_void_p * 2)(*[im.handle for im in imgs])
    pairs = np.array([[0, 1]], np.int32)
    ng, ni, npg = np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
    R, t = np.zeros(9), np.zeros(3)
    pf = matcher._match_params(None, 0.7, matcher.MODEL_FUNDAMENTAL, matcher.RANSAC_FIXED, 0.99, 1.0, 100, 64)
    rc = L.sfm_gpu_match_pairs_pose(gpu_ctx.handle, arr, 2, matcher.ptr(pairs), 1, ctypes.byref(pf), matcher.ptr(ng),
                                    matcher.ptr(ni), None, 0, 50.0, None, matcher.ptr(R), matcher.ptr(t),
                                    matcher.ptr(npg))
    assert rc == _lib.ERR_ARG   # no pose from a fundamental matrix
    pe = matcher._match_params(synth.REF_K, 0.7, matcher.MODEL_ESSENTIAL, matcher.RANSAC_FIXED, 0.99, 1.0, 100, 64)
    rc = L.sfm_gpu_match_pairs_pose(gpu_ctx.handle, arr, 2, matcher.ptr(pairs), 1, ctypes.byref(pe), matcher.ptr(ng),
                                    matcher.ptr(ni), None, 0, 50.0, None, None, matcher.ptr(t), matcher.ptr(npg))
    assert rc == _lib.ERR_ARG
    # identical images: every row matches itself at distance 0 and fails the ratio test or not, either way it runs
    rc = L.sfm_gpu_match_pairs_pose(gpu_ctx.handle, arr, 2, matcher.ptr(pairs), 1, ctypes.byref(pe), matcher.ptr(ng),
                                    matcher.ptr(ni), None, 0, 50.0, None, matcher.ptr(R), matcher.ptr(t),
                                    matcher.ptr(npg))
    assert rc == 0
    for im in imgs:
        im.close()


def _lists(rng, n_kp, sizes, dup):
    out = []
    for n in sizes:
        m = np.zeros(n, dtype=matcher.DMATCH_DTYPE)
        q = np.sort(rng.choice(n_kp, n, replace=False))
        for _ in range(dup if n > 1 else 0):
            i = int(rng.integers(1, n))
            q[i] = q[i - 1]
        m["queryIdx"] = np.sort(q)
        m["trainIdx"] = rng.choice(n_kp, n, replace=False)
        m["distance"] = rng.uniform(0, 300, n)
        out.append(m)
    return out


@pytest.mark.parametrize("n_kp,sizes,dup", [(3000, [900, 1100, 0, 700, 1024, 1025, 1], 0),
                                            (20000, [9000, 12000, 11000, 10500], 25),
                                            (500, [500, 500], 0), (64, [1, 1, 1], 0)])
def test_track_join_sequence_equals_the_loop_per_image(gpu_ctx, oracle, n_kp, sizes, dup):
    rng = np.random.default_rng(n_kp + len(sizes))
    ls = _lists(rng, n_kp, sizes, dup)
    got = matcher.track_join_sequence(gpu_ctx, ls)
    assert len(got) == len(sizes) - 1
    total = 0
    for l in range(len(sizes) - 1):
        want = oracle.track_join(ls[l], ls[l + 1])
        assert got[l].shape == want.shape and (got[l] == want).all()
        total += len(want)
    if max(sizes) >= 900:
        assert total > 0


def test_track_join_sequence_after_match_pairs(gpu_ctx, oracle):
    descs, kps = _ring(4, 2048, 800, 77)
    imgs = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    pairs = np.array([[0, 1], [1, 2], [2, 3]], np.int32)
    _, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, synth.REF_K, mode=matcher.RANSAC_FIXED, n_hyp=128,
                                        prob=0.99, threshold=1.0, want_matches=True)
    assert (n_inl > 300).all()
    got = matcher.track_join_sequence(gpu_ctx, sel)
    for l in range(2):
        want = oracle.track_join(sel[l], sel[l + 1])
        assert (got[l] == want).all() and got[l].shape == want.shape
        assert (got[l] == matcher.track_join(gpu_ctx, sel[l], sel[l + 1])).all()
    for im in imgs:
        im.close()


def test_track_join_sequence_limits(gpu_ctx):
    L = gpu_ctx._L
    rng = np.random.default_rng(9)
    ls = _lists(rng, 1000, [400, 500, 450], 0)
    want = matcher.track_join_sequence(gpu_ctx, ls)
    total = sum(len(w) for w in want)
    assert total > 10
    packed = np.zeros((3, 500), dtype=matcher.DMATCH_DTYPE)
    for l, m in enumerate(ls):
        packed[l, :len(m)] = m
    sizes = np.array([400, 500, 450], np.int32)
    out = np.full((4, 3), -7, np.int32)
    offs = np.zeros(3, np.int32)
    # too little room: the offsets say how much is needed, nothing is written
    rc = L.sfm_gpu_track_join_sequence(gpu_ctx.handle, matcher.ptr(packed), matcher.ptr(sizes), 500, 3,
                                       matcher.ptr(out), 4, matcher.ptr(offs))
    assert rc == _lib.ERR_CAPACITY and offs[2] == total and offs[1] == len(want[0]) and (out == -7).all()
    # a second list out of order is refused
    bad = packed.copy()
    bad[1, :2] = bad[1, 1::-1]
    assert bad[1, 0]["queryIdx"] > bad[1, 1]["queryIdx"]
    big = np.zeros((2000, 3), np.int32)
    rc = L.sfm_gpu_track_join_sequence(gpu_ctx.handle, matcher.ptr(bad), matcher.ptr(sizes), 500, 3,
                                       matcher.ptr(big), 2000, matcher.ptr(offs))
    assert rc == _lib.ERR_UNSUPPORTED
    # the first list may be in any order
    first = packed.copy()
    first[0, :400] = first[0, rng.permutation(400)]
    rc = L.sfm_gpu_track_join_sequence(gpu_ctx.handle, matcher.ptr(first), matcher.ptr(sizes), 500, 3,
                                       matcher.ptr(big), 2000, matcher.ptr(offs))
    assert rc == 0 and offs[2] == total
    # fewer than two lists: nothing to join; a size beyond the capacity: argument error
    assert matcher.track_join_sequence(gpu_ctx, []) == [] and matcher.track_join_sequence(gpu_ctx, [ls[0]]) == []
    sizes_bad = np.array([400, 501, 450], np.int32)
    rc = L.sfm_gpu_track_join_sequence(gpu_ctx.handle, matcher.ptr(packed), matcher.ptr(sizes_bad), 500, 3,
                                       matcher.ptr(big), 2000, matcher.ptr(offs))
    assert rc == _lib.ERR_ARG


def test_match_pairs_pose_over_two_launch_groups(gpu_ctx):
    """More pairs than one launch group holds (256): the poses of the second
    group land at their own places and equal one-pair calls."""
    rng = np.random.default_rng(32)
    pool = synth.sift_like(600, 52)
    images = []
    for v in range(12):
        n = 300 + 17 * v
        d = np.clip(pool[rng.permutation(600)[:n]] + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.float32)
        kp = np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
        images.append(matcher.Image(gpu_ctx, d, kp))
    pairs = np.array([[i, j] for i in range(12) for j in range(12) if i != j] * 2, np.int32)[:261]
    kw = dict(ratio=0.8, mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=32)
    res = matcher.match_pairs_pose(gpu_ctx, images, pairs, synth.REF_K, **kw)
    assert (res["n_pose_good"] >= 0).sum() > 200
    for p in (0, 100, 255, 256, 260):
        one = matcher.match_pairs_pose(gpu_ctx, images, pairs[p:p + 1], synth.REF_K, **kw)
        for key in ("n_good", "n_inliers", "n_pose_good"):
            assert one[key][0] == res[key][p]
        for key in ("E", "R", "t"):
            assert one[key][0].tobytes() == res[key][p].tobytes()
    for im in images:
        im.close()
