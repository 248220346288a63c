"""GPU three-view track join (sfm_gpu_track_join, the nested loop of
main.cpp:442-456) against the CPU oracle's literal O(m n) loop: identical
triples in identical order."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def lists(rng, n_kp, n_ab, n_bc, dup=0, dtype=None):
    ab = np.zeros(n_ab, dtype=dtype)
    ab["queryIdx"] = np.sort(rng.choice(n_kp, n_ab, replace=False))
    ab["trainIdx"] = rng.choice(n_kp, n_ab, replace=False)
    ab["distance"] = rng.uniform(0, 300, n_ab)
    bc = np.zeros(n_bc, dtype=dtype)
    q = np.sort(rng.choice(n_kp, n_bc, replace=False))
    for _ in range(dup):
        i = int(rng.integers(1, n_bc))
        q[i] = q[i - 1]
    bc["queryIdx"] = np.sort(q)
    bc["trainIdx"] = rng.choice(n_kp, n_bc, replace=False)
    return ab, bc


@pytest.mark.parametrize("n_kp,n_ab,n_bc,dup", [(50, 1, 1, 0), (2000, 700, 650, 0), (2000, 1023, 1024, 0),
                                                (2000, 1025, 1500, 5), (8192, 3100, 3300, 0),
                                                (32768, 13000, 12500, 40), (300, 300, 300, 0)])
def test_track_join_matches_the_loop(gpu_ctx, oracle, n_kp, n_ab, n_bc, dup):
    from sfm_amd import matcher
    rng = np.random.default_rng(n_kp + n_ab + dup)
    ab, bc = lists(rng, n_kp, n_ab, n_bc, dup, matcher.DMATCH_DTYPE)
    got = matcher.track_join(gpu_ctx, ab, bc)
    want = oracle.track_join(ab, bc)
    assert got.shape == want.shape and (got == want).all()
    if n_ab >= 700:
        assert len(want) > 0
    # the first list may come in any order: the output follows it
    perm = rng.permutation(n_ab)
    got = matcher.track_join(gpu_ctx, ab[perm], bc)
    want = oracle.track_join(ab[perm], bc)
    assert (got == want).all()


def test_track_join_after_match_pairs(gpu_ctx, oracle):
    """The lists as the library itself emits them for a ring of three images."""
    from sfm_amd import matcher, synth
    n = 2048
    Q, T, planted = synth.planted_pair(n, n, 51, 52, 53)
    T2 = T.copy()
    rng = np.random.default_rng(54)
    T2 = T2[rng.permutation(n)]          # third image: the second one's rows shuffled
    imgs = [matcher.Image(gpu_ctx, d, rng.uniform(0, 3000, (n, 2)).astype(np.float32)) for d in (Q, T, T2)]
    pairs = np.array([[0, 1], [1, 2]], np.int32)
    n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, None, ratio=0.7, model=matcher.MODEL_FUNDAMENTAL,
                                             mode=matcher.RANSAC_FIXED, prob=0.99, threshold=1e6, n_hyp=64,
                                             want_matches=True)
    ab, bc = sel[0], sel[1]
    assert n_inl[0] > 100 and n_inl[1] > 100
    assert (np.diff(bc["queryIdx"]) > 0).all()      # what the join relies on
    got = matcher.track_join(gpu_ctx, ab, bc)
    want = oracle.track_join(ab, bc)
    assert len(want) > 50 and (got == want).all()
    for im in imgs:
        im.close()


def test_track_join_limits(gpu_ctx):
    from sfm_amd import matcher
    L = gpu_ctx._L
    rng = np.random.default_rng(1)
    ab, bc = lists(rng, 500, 200, 200, 0, matcher.DMATCH_DTYPE)
    assert matcher.track_join(gpu_ctx, ab[:0], bc).shape == (0, 3)
    assert matcher.track_join(gpu_ctx, ab, bc[:0]).shape == (0, 3)
    # second list out of order: refused, not silently wrong
    bad = bc[::-1].copy()
    with pytest.raises(matcher.SfmGpuError) as ei:
        matcher.track_join(gpu_ctx, ab, bad)
    assert "ascending" in str(ei.value)
    # capacity too small: the count comes back, no triples are written
    full = matcher.track_join(gpu_ctx, ab, bc)
    assert len(full) > 3
    out = np.full((2, 3), -7, np.int32)
    n = ctypes.c_int32(0)
    rc = L.sfm_gpu_track_join(gpu_ctx.handle, ab.ctypes.data, len(ab), bc.ctypes.data, len(bc), out.ctypes.data, 2,
                              ctypes.byref(n))
    assert rc == -7 and n.value == len(full) and (out == -7).all()
    assert L.sfm_gpu_track_join(gpu_ctx.handle, None, 5, bc.ctypes.data, len(bc), out.ctypes.data, 2,
                                ctypes.byref(n)) == -1
    assert (matcher.track_join(gpu_ctx, ab, bc) == full).all()


def test_sequence_from_blobs_gives_world_tracks(gpu_ctx, oracle, tmp_path):
    """The chain the reference runs (match() ring, then the three-view join) from
    descriptor blobs on disk: the tracks found are the planted world points."""
    from sfm_amd import blob, matcher, synth
    descs, kps, ids = synth.image_sequence(4, 1500, 9)
    for i in range(4):
        blob.write_blob(str(tmp_path / ("img_%d.sfmd" % i)), descs[i].astype(np.uint8) if i % 2 else descs[i],
                        kps[i], synth.REF_K)
    blobs = [blob.read_blob(str(tmp_path / ("img_%d.sfmd" % i))) for i in range(4)]
    imgs = [matcher.Image(gpu_ctx, np.ascontiguousarray(b["descriptors"]), np.ascontiguousarray(b["keypoints"]))
            for b in blobs]
    pairs = np.array([[0, 1], [1, 2], [2, 3]], np.int32)
    n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, blobs[0]["K"], prob=0.99, threshold=1.0,
                                             want_matches=True)
    for p, (i, j) in enumerate(pairs.tolist()):
        # the same pair through the oracle, call by call
        m, c = oracle.knn_match(descs[i], descs[j], 2)
        keep = oracle.ratio_test(m, c, 0.7)
        good = m[keep, 0]
        r = oracle.ransac(kps[i][good["queryIdx"]].astype(np.float64), kps[j][good["trainIdx"]].astype(np.float64),
                          synth.REF_K, prob=0.99, threshold=1.0)
        want = good[r["mask"] != 0]
        assert n_good[p] == len(good) and n_inl[p] == len(want)
        assert (sel[p]["queryIdx"] == want["queryIdx"]).all() and (sel[p]["trainIdx"] == want["trainIdx"]).all()
    for p in range(2):
        tr = matcher.track_join(gpu_ctx, sel[p], sel[p + 1])
        assert (tr == oracle.track_join(sel[p], sel[p + 1])).all() and len(tr) > 100
        a, b, c = ids[p][tr[:, 0]], ids[p + 1][tr[:, 1]], ids[p + 2][tr[:, 2]]
        assert (a == b).all() and (b == c).all() and (a >= 0).all()
    for im in imgs:
        im.close()
