"""GPU parity of the RANSAC path against the CPU oracle, through the C ABI.

Bar: the sample table, the per-hypothesis inlier counts and the inlier mask are
bit-exact; the model agrees to 1e-6 relative (in practice to the last bit,
which the tests report).
"""
import ctypes

import numpy as np
import pytest

from sfm_amd import matcher, synth
from sfm_amd._lib import ptr

pytestmark = pytest.mark.gpu


def _dump(ctx, n_hyp, m):
    table = np.zeros((n_hyp, m), np.int32)
    nmodels = np.zeros(n_hyp, np.int32)
    counts = np.zeros((n_hyp, 10), np.int32)
    models = np.zeros((n_hyp, 10, 3, 3), np.float64)
    ctx.check(ctx._L.sfm_gpu_debug_ransac_dump(ctx.handle, n_hyp, ptr(table), ptr(nmodels),
                                               ptr(counts), ptr(models)))
    return table, nmodels, counts, models


@pytest.fixture
def full_counts(gpu_ctx):
    """Fixed-count scoring normally drops models that cannot win between its
    stages; tests that compare every hypothesis' count switch that off."""
    gpu_ctx._L.sfm_gpu_debug_set_score_prune(gpu_ctx.handle, 0)
    yield
    gpu_ctx._L.sfm_gpu_debug_set_score_prune(gpu_ctx.handle, 1)


def _model_close(a, b, tol=1e-6):
    return np.abs(a - b).max() <= tol * np.abs(b).max()


@pytest.mark.parametrize("seed,n,frac", [(7, 3000, 0.9), (8, 600, 0.6), (9, 14000, 0.8)])
def test_essential_fixed_counts_mask_model(gpu_ctx, oracle, full_counts, seed, n, frac):
    p1, p2, is_in, _ = synth.two_view(n, seed, inlier_frac=frac)
    nh = 256
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99, threshold=1.0,
                                          mode=matcher.RANSAC_FIXED, n_hyp=nh)
    o = oracle.ransac(p1, p2, synth.REF_K, prob=0.99, threshold=1.0, fixed_count=nh,
                      want_counts=True)
    table, nmodels, counts, models = _dump(gpu_ctx, nh, 5)
    assert (table == oracle.sample_table_fixed(n, 5, nh)).all()
    assert (counts == o["counts"]).all()
    assert r["ok"] == o["ok"] == 1
    assert r["n_inliers"] == o["n_inliers"] and r["iters"] == nh
    assert (mask == o["mask"]).all()
    assert _model_close(E, o["model"])
    assert (E == o["model"]).all(), "model differs in the last bits"


def test_five_point_models_bit_exact(gpu_ctx, oracle):
    p1, p2, _, _ = synth.two_view(500, 3, inlier_frac=0.7)
    nh = 128
    matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, mode=matcher.RANSAC_FIXED, n_hyp=nh)
    table, nmodels, counts, models = _dump(gpu_ctx, nh, 5)
    x1, _ = oracle.normalize_points(p1, synth.REF_K, 1.0)
    x2, _ = oracle.normalize_points(p2, synth.REF_K, 1.0)
    total = 0
    for h in range(nh):
        om = oracle.five_point(x1[table[h]], x2[table[h]])
        assert len(om) == nmodels[h], h
        assert (models[h, :len(om)] == om).all(), h
        total += len(om)
    assert total > nh      # on average more than one candidate per hypothesis


def test_essential_custom_sample_table(gpu_ctx, oracle):
    p1, p2, _, _ = synth.two_view(2000, 5, inlier_frac=0.75)
    table = oracle.sample_table(2000, 5, 300)      # the rows OpenCV's RNG(-1) draws
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99,
                                          mode=matcher.RANSAC_FIXED, sample_table=table)
    o = oracle.ransac(p1, p2, synth.REF_K, prob=0.99, fixed_count=300, sample_table=table)
    assert r["ok"] == 1 and (mask == o["mask"]).all() and r["n_inliers"] == o["n_inliers"]
    assert _model_close(E, o["model"])


@pytest.mark.parametrize("seed,n,frac", [(11, 1500, 0.85), (12, 4000, 0.91), (13, 700, 0.45),
                                         (14, 300, 0.25)])
def test_essential_adaptive_matches_sequential_loop(gpu_ctx, oracle, seed, n, frac):
    p1, p2, _, _ = synth.two_view(n, seed, inlier_frac=frac, noise_px=0.4)
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99, threshold=1.0)
    o = oracle.ransac(p1, p2, synth.REF_K, prob=0.99, threshold=1.0)
    assert r["ok"] == o["ok"]
    assert r["iters"] == o["iters"]
    assert r["n_inliers"] == o["n_inliers"]
    assert (mask == o["mask"]).all()
    if o["ok"]:
        assert _model_close(E, o["model"])


def test_essential_degenerate_sizes(gpu_ctx, oracle):
    p1, p2, _, _ = synth.two_view(5, 1, inlier_frac=1.0, noise_px=0.0)
    sentinel = np.full(4, 7, np.uint8)
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1[:4], p2[:4], synth.REF_K, mask=sentinel)
    assert E is None and mask.tolist() == [7, 7, 7, 7]
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99)
    o = oracle.ransac(p1, p2, synth.REF_K, prob=0.99)
    assert r["ok"] == o["ok"] == 1 and mask.tolist() == [1] * 5
    assert _model_close(E, o["model"])
    # collinear / repeated points: no crash, same verdict as the oracle
    q1 = np.tile(p1[:1], (40, 1))
    q2 = np.tile(p2[:1], (40, 1))
    E, mask, r = matcher.findEssentialMat(gpu_ctx, q1, q2, synth.REF_K, prob=0.99, maxIters=50)
    o = oracle.ransac(q1, q2, synth.REF_K, prob=0.99, max_iters=50)
    assert r["ok"] == o["ok"] and (mask == o["mask"]).all()


@pytest.fixture(params=[0, 1], ids=["one_lane_per_hypothesis", "two_lanes_per_hypothesis"])
def solver_split(request, gpu_ctx):
    """Both forms of the seven-point elimination, whatever the batch size would pick."""
    gpu_ctx.check(gpu_ctx._L.sfm_gpu_debug_set_solver_split(gpu_ctx.handle, request.param))
    yield request.param
    gpu_ctx.check(gpu_ctx._L.sfm_gpu_debug_set_solver_split(gpu_ctx.handle, -1))


@pytest.mark.parametrize("seed,n", [(21, 3300), (22, 900), (23, 40)])
def test_fundamental_fixed_and_adaptive(gpu_ctx, oracle, full_counts, solver_split, seed, n):
    p1, p2, _, _ = synth.two_view(n, seed, inlier_frac=0.8)
    nh = 512
    F, mask, r = matcher.findFundamentalMat(gpu_ctx, p1, p2, ransacReprojThreshold=3.0,
                                            confidence=0.99, mode=matcher.RANSAC_FIXED, n_hyp=nh)
    o = oracle.ransac(p1, p2, None, model=1, prob=0.99, threshold=3.0, fixed_count=nh,
                      want_counts=True)
    table, nmodels, counts, models = _dump(gpu_ctx, nh, 7)
    assert (table == oracle.sample_table_fixed(n, 7, nh, p1, p2)).all()
    assert (counts == o["counts"]).all()
    assert r["ok"] == 1 and (mask == o["mask"]).all() and r["n_inliers"] == o["n_inliers"]
    assert _model_close(F, o["model"])
    F, mask, r = matcher.findFundamentalMat(gpu_ctx, p1, p2, ransacReprojThreshold=3.0,
                                            confidence=0.99)
    o = oracle.ransac(p1, p2, None, model=1, prob=0.99, threshold=3.0)
    assert r["ok"] == o["ok"] == 1 and r["iters"] == o["iters"]
    assert (mask == o["mask"]).all()
    assert _model_close(F, o["model"])


def test_full_size_4096_hyps_properties(gpu_ctx):
    # full-size fixed-count run: the mask is reproducible, consistent with the
    # returned model, and the inlier count is the maximum over what adaptive finds
    p1, p2, is_in, E_true = synth.two_view(13700, 2101, inlier_frac=0.9)
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99,
                                          mode=matcher.RANSAC_FIXED, n_hyp=4096)
    E2, mask2, r2 = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99,
                                             mode=matcher.RANSAC_FIXED, n_hyp=4096)
    assert (mask == mask2).all() and (E == E2).all()
    assert r["n_inliers"] == int(mask.sum())
    Ea, maska, ra = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99)
    assert r["n_inliers"] >= ra["n_inliers"]
    assert (mask.astype(bool) & ~is_in).sum() < 30
    assert abs(np.linalg.det(E)) < 1e-10
    assert min(np.abs(E - E_true).max(), np.abs(E + E_true).max()) < 0.02


def _near_threshold_problem(model, seed, n_clean, n_edge, thr_px):
    """Correspondences of one true geometry: `n_clean` exact ones and `n_edge`
    whose second point is pushed off its epipolar line by thr*(1 +- 2 %), so
    that for hypotheses drawn from the clean ones a large share of the tests
    lands inside the band the float32 filter cannot settle."""
    rng = np.random.default_rng(seed)
    p1, p2, _, E = synth.two_view(n_clean + n_edge, seed, inlier_frac=1.0, noise_px=0.0)
    K = synth.REF_K
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ E @ Kinv
    h1 = np.concatenate([p1, np.ones((len(p1), 1))], 1)
    l2 = h1 @ F.T                                   # epipolar lines in image 2
    nrm = l2[:, :2] / np.linalg.norm(l2[:, :2], axis=1, keepdims=True)
    # the Sampson error of a one-sided displacement d is about d^2 / 2
    scale = np.sqrt(2.0) if model == 0 else 1.0
    delta = scale * thr_px * (1.0 + rng.uniform(-0.02, 0.02, len(p1))) * rng.choice([-1.0, 1.0], len(p1))
    delta[:n_clean] = 0.0
    p2 = p2 + nrm * delta[:, None]
    p1 = p1.astype(np.float32).astype(np.float64)
    p2 = p2.astype(np.float32).astype(np.float64)
    m = 5 if model == 0 else 7
    table = np.stack([rng.permutation(n_clean)[:m] for _ in range(192)]).astype(np.int32)
    return p1, p2, table


@pytest.mark.parametrize("model,coord_scale", [(0, 1.0), (1, 1.0), (1, 64.0), (1, 1.0 / 64.0)])
def test_scoring_filter_is_exact_on_near_threshold_points(gpu_ctx, oracle, full_counts, model, coord_scale):
    thr = 1.0 if model == 0 else 3.0
    p1, p2, table = _near_threshold_problem(model, 77 + model, 1500, 4500, thr)
    if coord_scale != 1.0:
        # the same geometry in other units (the filter's error bound scales with
        # the largest coordinate): pixels and threshold scaled alike
        p1 = (p1 * coord_scale).astype(np.float32).astype(np.float64)
        p2 = (p2 * coord_scale).astype(np.float32).astype(np.float64)
        thr = thr * coord_scale
    K = synth.REF_K if model == 0 else None
    o = oracle.ransac(p1, p2, K, model=model, prob=0.99, threshold=thr, fixed_count=len(table),
                      sample_table=table, want_counts=True)
    res = {}
    for filt in (1, 0):
        gpu_ctx._L.sfm_gpu_debug_set_score_filter(gpu_ctx.handle, filt)
        try:
            if model == 0:
                _, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, K, prob=0.99, threshold=thr,
                                                      mode=matcher.RANSAC_FIXED, sample_table=table)
            else:
                _, mask, r = matcher.findFundamentalMat(gpu_ctx, p1, p2, ransacReprojThreshold=thr,
                                                        confidence=0.99, mode=matcher.RANSAC_FIXED,
                                                        sample_table=table)
        finally:
            gpu_ctx._L.sfm_gpu_debug_set_score_filter(gpu_ctx.handle, 1)
        _, _, counts, _ = _dump(gpu_ctx, len(table), 5 if model == 0 else 7)
        res[filt] = (counts.copy(), mask.copy(), r["n_inliers"])
        assert (counts == o["counts"]).all()
        assert (mask == o["mask"]).all() and r["n_inliers"] == o["n_inliers"]
    assert (res[0][0] == res[1][0]).all()
    # the construction did its job: the best model splits the edge points, neither
    # all in nor all out, so many tests sat near the threshold
    best = o["counts"].max()
    assert 1500 + 200 < best < 6000 - 200


@pytest.mark.parametrize("poison", ["big", "huge", "inf", "nan"])
@pytest.mark.parametrize("model", [0, 1])
def test_scoring_with_huge_or_non_finite_coordinates(gpu_ctx, oracle, full_counts, model, poison):
    """A few correspondences whose coordinates are huge, infinite or not numbers
    (none of them in a sample): the float32 filter reads its verdicts off sign
    bits and must not be asked about such a problem; every count, the mask and
    the winner are those of the double-precision loop."""
    rng = np.random.default_rng(5)
    p1, p2, _, _ = synth.two_view(2600, 91 + model, inlier_frac=0.8)
    p1 = p1.astype(np.float32).astype(np.float64)
    p2 = p2.astype(np.float32).astype(np.float64)
    # "big" still passes for a coordinate the filter may see (< 1e18); its models' terms
    # could overflow float32, so the filter refuses every model instead
    bad = {"big": 1e17, "huge": 3e30, "inf": np.inf, "nan": np.nan}[poison]
    rows = rng.permutation(np.arange(2000, 2600))[:24]
    p1[rows[:8], 0] = bad
    p2[rows[8:16], 1] = -bad
    p1[rows[16:], 1] = bad
    p2[rows[16:], 0] = bad
    m = 5 if model == 0 else 7
    thr = 1.0 if model == 0 else 3.0
    K = synth.REF_K if model == 0 else None
    table = np.stack([rng.permutation(2000)[:m] for _ in range(96)]).astype(np.int32)
    o = oracle.ransac(p1, p2, K, model=model, prob=0.99, threshold=thr, fixed_count=len(table),
                      sample_table=table, want_counts=True)
    if model == 0:
        M, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, K, prob=0.99, threshold=thr,
                                              mode=matcher.RANSAC_FIXED, sample_table=table)
    else:
        M, mask, r = matcher.findFundamentalMat(gpu_ctx, p1, p2, ransacReprojThreshold=thr, confidence=0.99,
                                                mode=matcher.RANSAC_FIXED, sample_table=table)
    _, _, counts, _ = _dump(gpu_ctx, len(table), m)
    assert (counts == o["counts"]).all()
    assert (mask == o["mask"]).all() and r["n_inliers"] == o["n_inliers"] and r["ok"] == o["ok"]
    assert o["n_inliers"] > 1000 and not mask[rows].any()


@pytest.mark.parametrize("prune_mode", [1, 2, 3, 4])
@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("seed,n,frac", [(31, 400, 0.9), (32, 513, 0.5), (33, 3300, 0.93), (34, 3300, 0.35),
                                         (35, 14000, 0.8), (36, 5000, 0.05), (37, 1100, 0.97), (38, 2047, 0.7)])
def test_pruned_scoring_selects_the_same_model(gpu_ctx, oracle, model, seed, n, frac, prune_mode):
    """Fixed-count scoring with models that cannot win dropped along the way (the
    default schedule, 1, and the two earlier ones) against the oracle's exhaustive
    loop: same winner, count and mask; the dumped counts are complete for every
    model that reaches the best count and never exceed the true count elsewhere."""
    p1, p2, _, _ = synth.two_view(n, seed, inlier_frac=frac)
    nh = 256
    m = 5 if model == 0 else 7
    thr = 1.0 if model == 0 else 3.0
    K = synth.REF_K if model == 0 else None
    o = oracle.ransac(p1, p2, K, model=model, prob=0.99, threshold=thr, fixed_count=nh, want_counts=True)
    gpu_ctx.check(gpu_ctx._L.sfm_gpu_debug_set_score_prune(gpu_ctx.handle, prune_mode))
    try:
        if model == 0:
            M, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, K, prob=0.99, threshold=thr,
                                                  mode=matcher.RANSAC_FIXED, n_hyp=nh)
        else:
            M, mask, r = matcher.findFundamentalMat(gpu_ctx, p1, p2, ransacReprojThreshold=thr, confidence=0.99,
                                                    mode=matcher.RANSAC_FIXED, n_hyp=nh)
    finally:
        gpu_ctx._L.sfm_gpu_debug_set_score_prune(gpu_ctx.handle, 1)
    _, _, counts, _ = _dump(gpu_ctx, nh, m)
    assert r["ok"] == o["ok"] and r["n_inliers"] == o["n_inliers"]
    assert (mask == o["mask"]).all()
    if o["ok"]:
        assert (np.asarray(M) == o["model"]).all()
    oc = o["counts"]
    assert (counts <= oc).all() and ((counts == -1) == (oc == -1)).all()
    top = oc == oc.max()
    assert (counts[top] == oc[top]).all()
    if n >= 3000 and frac >= 0.9:
        # the first stretch outnumbers the outliers, so the bound bites: most
        # models were left with partial counts
        assert (counts < oc)[oc >= 0].mean() > 0.3


def test_stage_sizes_chosen_on_the_device_never_change_the_result(gpu_ctx, oracle):
    """The scoring stages are sized per problem on the device from a first group
    of models scored on every point; calls with very different outlier shares in
    any order give the exhaustive loop's answer each time."""
    pa1, pa2, _, _ = synth.two_view(6000, 61, inlier_frac=0.95)
    pb1, pb2, _, _ = synth.two_view(6000, 62, inlier_frac=0.3)
    want = {}
    for name, (p1, p2) in (("a", (pa1, pa2)), ("b", (pb1, pb2))):
        o = oracle.ransac(p1, p2, None, model=1, prob=0.99, threshold=3.0, fixed_count=192)
        want[name] = (o["mask"], o["model"], o["n_inliers"])
    for name in ("a", "b", "a", "a", "b", "b", "a"):   # nothing carries over between calls
        p1, p2 = (pa1, pa2) if name == "a" else (pb1, pb2)
        F, mask, r = matcher.findFundamentalMat(gpu_ctx, p1, p2, ransacReprojThreshold=3.0, confidence=0.99,
                                                mode=matcher.RANSAC_FIXED, n_hyp=192)
        assert (mask == want[name][0]).all() and r["n_inliers"] == want[name][2]
        assert (np.asarray(F) == want[name][1]).all()


def test_minimal_point_counts_return_every_candidate(gpu_ctx, oracle):
    """n == 5 (essential) and n == 7 (fundamental): OpenCV runs no RANSAC and
    returns every candidate of the one direct solve stacked; the first goes to
    the model output, sfm_gpu_ransac_candidates delivers the stack."""
    for seed in (1, 2, 3):
        p1, p2, _, _ = synth.two_view(7, seed, inlier_frac=1.0, noise_px=0.3)
        E, mask, r = matcher.findEssentialMat(gpu_ctx, p1[:5], p2[:5], synth.REF_K, prob=0.99)
        o = oracle.ransac(p1[:5], p2[:5], synth.REF_K, prob=0.99)
        want = oracle.ransac_direct_models()
        got = matcher.ransac_candidates(gpu_ctx)
        assert r["ok"] == o["ok"] == 1 and mask.tolist() == [1] * 5
        assert got.shape == want.shape and got.shape[0] >= 1
        assert (got == want).all() and (got[0] == np.asarray(E)).all()
        F, mask, r = matcher.findFundamentalMat(gpu_ctx, p1, p2, ransacReprojThreshold=3.0)
        o = oracle.ransac(p1, p2, None, model=1, prob=0.99, threshold=3.0)
        want = oracle.ransac_direct_models()
        got = matcher.ransac_candidates(gpu_ctx)
        assert r["ok"] == o["ok"] == 1 and mask.tolist() == [1] * 7
        assert got.shape == want.shape and 1 <= got.shape[0] <= 3
        assert (got == want).all() and (got[0] == np.asarray(F)).all()
    # after an ordinary run: the candidates of the winning sample, the model among them
    p1, p2, _, _ = synth.two_view(400, 9, inlier_frac=0.8)
    E, mask, r = matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, prob=0.99)
    got = matcher.ransac_candidates(gpu_ctx)
    assert any((g == np.asarray(E)).all() for g in got)


def test_fundamental_between_8_and_14_points_is_refused(gpu_ctx):
    """cv::findFundamentalMat(FM_RANSAC) uses LMedS for 8..14 points: not this path."""
    from sfm_amd import _lib
    p1, p2, _, _ = synth.two_view(20, 4, inlier_frac=1.0)
    for n in (8, 11, 14):
        with pytest.raises(matcher.SfmGpuError) as e:
            matcher.findFundamentalMat(gpu_ctx, p1[:n], p2[:n])
        assert e.value.code == _lib.ERR_UNSUPPORTED
        # the fixed-count benchmark mode has no OpenCV counterpart and takes any n >= 7
        F, mask, r = matcher.findFundamentalMat(gpu_ctx, p1[:n], p2[:n], mode=matcher.RANSAC_FIXED, n_hyp=16)
        assert r["ok"] in (0, 1)
    F, mask, r = matcher.findFundamentalMat(gpu_ctx, p1[:15], p2[:15])
    assert r["ok"] == 1


def test_bad_parameters_are_rejected(gpu_ctx):
    from sfm_amd import _lib
    p1, p2, _, _ = synth.two_view(50, 5, inlier_frac=1.0)
    K0 = synth.REF_K.copy()
    K0[0, 0] = K0[1, 1] = 0.0
    bad = [dict(threshold=0.0), dict(threshold=-1.0), dict(threshold=float("nan")), dict(threshold=float("inf")),
           dict(prob=1.0), dict(prob=float("nan"))]
    for kw in bad:
        with pytest.raises(matcher.SfmGpuError) as e:
            matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, **kw)
        assert e.value.code == _lib.ERR_ARG, kw
    with pytest.raises(matcher.SfmGpuError) as e:
        matcher.findEssentialMat(gpu_ctx, p1, p2, K0)
    assert e.value.code == _lib.ERR_ARG
    with pytest.raises(matcher.SfmGpuError) as e:
        matcher.findEssentialMat(gpu_ctx, p1, p2, synth.REF_K, mode=matcher.RANSAC_FIXED, n_hyp=(1 << 20) + 1)
    assert e.value.code == _lib.ERR_CAPACITY
