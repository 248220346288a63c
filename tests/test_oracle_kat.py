"""Known-answer tests that pin the CPU oracle without OpenCV.

The reference holds no tests or golden vectors for this path (parity
unpinned); these are the answers that can be derived independently: exact
integer distances, OpenCV's documented tie order, the cv::RNG recurrence, the
RANSAC iteration-count formula and the algebraic properties of an essential
matrix.
"""
import numpy as np
import pytest

from sfm_amd import synth


def test_knn_hand_rows(oracle):
    t = np.zeros((3, 128), np.float32)
    q = np.zeros((1, 128), np.float32)
    t[1, 5] = 1      # d2 = 1
    t[2, 7] = 2      # d2 = 4
    m, c = oracle.knn_match(q, t, 2)
    assert c[0] == 2
    assert m["trainIdx"][0].tolist() == [0, 1]
    assert m["distance"][0].tolist() == [0.0, 1.0]
    m, c = oracle.knn_match(q, t[1:], 2)
    assert m["trainIdx"][0].tolist() == [0, 1] and m["distance"][0].tolist() == [1.0, 2.0]


def test_knn_duplicates_keep_lowest_index(oracle):
    rng = np.random.default_rng(0)
    t = rng.integers(0, 256, (40, 128)).astype(np.float32)
    t[31] = t[3]
    t[17] = t[3]
    q = t[3:4].copy()
    m, c = oracle.knn_match(q, t, 2)
    assert m["trainIdx"][0].tolist() == [3, 17]
    assert m["distance"][0].tolist() == [0.0, 0.0]


def test_knn_short_rows_and_empty(oracle):
    q = np.zeros((2, 128), np.float32)
    t = np.full((1, 128), 255, np.float32)
    m, c = oracle.knn_match(q, t, 2)
    assert c.tolist() == [1, 1]
    assert m["trainIdx"][:, 0].tolist() == [0, 0] and m["trainIdx"][:, 1].tolist() == [-1, -1]
    # all-0 vs all-255: d2 = 128*255^2 = 8323200
    assert m["distance"][0, 0] == np.float32(np.sqrt(np.float64(8323200.0)))
    m, c = oracle.knn_match(q, np.zeros((0, 128), np.float32), 2)
    assert c.tolist() == [0, 0]


def test_knn_float_collision_prefers_lower_index(oracle):
    # two different d2 >= 2^22 with the same float32 sqrt: the ranking is on the
    # float, so the lower train index wins although its d2 is larger
    base = None
    for d2 in range(1 << 22, (1 << 22) + 4000):
        if np.float32(np.sqrt(np.float64(d2))) == np.float32(np.sqrt(np.float64(d2 + 1))):
            base = d2
            break
    assert base is not None
    q = np.zeros((1, 128), np.float32)

    def row_with_d2(target):
        r = np.zeros(128, np.float32)
        rem = target
        for i in range(128):
            v = min(255, int(np.floor(np.sqrt(rem))))
            r[i] = v
            rem -= v * v
        assert rem == 0, rem
        return r
    t = np.stack([row_with_d2(base + 1), row_with_d2(base), np.full(128, 255, np.float32)])
    m, c = oracle.knn_match(q, t, 2)
    assert m["trainIdx"][0].tolist() == [0, 1]
    assert m["distance"][0, 0] == m["distance"][0, 1]


def test_ratio_test_in_double(oracle):
    m = np.zeros((3, 2), dtype=oracle.DMATCH_DTYPE)
    c = np.array([2, 2, 1], np.int32)
    # 0.7f * 10 vs 0.7 * 10: pick d0 so that float and double products disagree
    d1 = np.float32(10.0)
    d0 = np.float32(0.7) * d1      # = 7.0000005 > 0.7*10 = 7.000000000000001 in double
    m["distance"][0] = (d0, d1)
    m["distance"][1] = (np.float32(6.9999995), d1)
    m["distance"][2] = (1.0, 0.0)
    keep = oracle.ratio_test(m, c, 0.7)
    assert keep.tolist() == [1]


def test_cv_rng_stream(oracle):
    assert oracle.rng_next(8).tolist() == [0x07c09cf5, 0xbac3439c, 0x99ae7b8c, 0x37275f45,
                                           0x40637cc7, 0xaf98a118, 0x059ae535, 0x8902c9a0]
    assert oracle.rng_uniform(0, 1000, 8).tolist() == [605, 4, 940, 173, 831, 512, 301, 280]
    assert oracle.rng_uniform(0, 8443, 8).tolist() == [7633, 1287, 271, 8145, 5310, 7965, 7610, 3872]


def test_sample_table_distinct_and_stream(oracle):
    tab = oracle.sample_table(8443, 5, 4)
    assert tab[0].tolist() == [7633, 1287, 271, 8145, 5310]
    assert tab[1].tolist()[:3] == [7965, 7610, 3872]
    small = oracle.sample_table(6, 5, 200)
    assert all(len(set(r.tolist())) == 5 for r in small)
    assert small.min() >= 0 and small.max() < 6


def test_fixed_table_rows_are_independent_and_distinct(oracle):
    a = oracle.sample_table_fixed(3000, 5, 64)
    b = oracle.sample_table_fixed(3000, 5, 2048)
    assert (a == b[:64]).all()                       # row r does not depend on the row count
    assert all(len(set(r)) == 5 for r in b.tolist())
    assert b.min() >= 0 and b.max() < 3000
    hist = np.bincount(b.ravel(), minlength=3000)
    assert hist.max() < 20                           # no gross bias
    small = oracle.sample_table_fixed(6, 5, 200)
    assert all(sorted(set(r)) == sorted(r) for r in small.tolist())


def test_real_roots_against_constructed_polynomials(oracle):
    rng = np.random.default_rng(0)
    for trial in range(600):
        deg = int(rng.integers(1, 11))
        nreal = int(rng.integers(0, deg + 1))
        roots = list(rng.normal(size=nreal) * rng.choice([0.1, 1, 10, 100]))
        p = np.poly1d([1.0])
        for x in roots:
            p *= np.poly1d([1, -x])
        while p.order < deg - 1:
            a, b = rng.normal(size=2)
            p *= np.poly1d([1, -2 * a, a * a + b * b + 0.01])
        if p.order < deg:
            x = rng.normal()
            p *= np.poly1d([1, -x])
            roots.append(x)
        got = oracle.real_roots(p.coeffs[::-1] * rng.normal() * 10)
        want = np.sort(roots)
        assert len(got) == len(want), (trial, want, got)
        if len(want):
            assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_real_roots_small_degrees(oracle):
    """Degrees <= 3 take the critical points from the quadratic formula.  (The
    coefficients are scaled by their largest magnitude first, so even integer
    roots come back only to a few units in the last place.)"""
    def check(coef, want):
        got = oracle.real_roots(coef)
        assert len(got) == len(want), (coef, got)
        for g, w in zip(got, want):
            assert abs(g - w) <= 1e-14 * max(1.0, abs(w)), (coef, got)
    check([2.0, -3.0, 1.0], [1.0, 2.0])            # (x-1)(x-2)
    check([1.0, 0.0, 1.0], [])                      # x^2 + 1
    check([1.0, -2.0, 1.0], [1.0])                  # (x-1)^2: the double root once
    check([2.0, -3.0, 1.0, 0.0], [1.0, 2.0])        # vanishing leading term
    check([-6.0, 11.0, -6.0, 1.0], [1.0, 2.0, 3.0])
    check([0.0, 0.0, 0.0, 1.0], [0.0])              # x^3
    check([1.0, 1.0, 1.0, 1.0], [-1.0])             # (x+1)(x^2+1)
    check([-8.0, 4.0], [2.0])
    # ill-scaled quadratic: the small root keeps full relative precision
    r = oracle.real_roots([1.0, -1e8, 1.0])
    assert abs(r[0] - 1e-8) < 1e-22 and abs(r[1] - 1e8) < 1e-6


def test_update_num_iters(oracle):
    assert oracle.update_num_iters(0.99, 1 - 0.91, 5, 1000) == 5
    assert oracle.update_num_iters(0.99, 0.5, 5, 1000) == 145
    assert oracle.update_num_iters(0.99, 1.0, 5, 1000) == 1000
    assert oracle.update_num_iters(0.99, 0.0, 5, 1000) == 0


def test_threshold_scaling(oracle):
    _, thr = oracle.normalize_points(np.zeros((1, 2)), synth.REF_K, 1.0)
    assert abs(thr - 3.4628976e-4) < 1e-10
    assert np.float32(thr * thr) == np.float32(1.199166e-07)


def _true_E_pair(seed, n=60):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, 3))
    X[:, 2] += 5
    R = synth._rot(rng.normal(size=3), 0.07)
    t = rng.normal(size=3)
    t /= np.linalg.norm(t)
    x1 = X[:, :2] / X[:, 2:]
    X2 = X @ R.T + t
    x2 = X2[:, :2] / X2[:, 2:]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return x1, x2, E / np.linalg.norm(E)


def test_five_point_recovers_true_E(oracle):
    for seed in range(10):
        x1, x2, E = _true_E_pair(seed)
        ms = oracle.five_point(x1[:5], x2[:5])
        assert 1 <= len(ms) <= 10
        best = min(min(np.abs(M - E).max(), np.abs(M + E).max()) for M in ms)
        assert best < 1e-8, (seed, best)
        for M in ms:
            assert abs(np.linalg.norm(M) - 1) < 1e-12
            assert abs(np.linalg.det(M)) < 1e-9
            assert np.abs(2 * M @ M.T @ M - np.trace(M @ M.T) * M).max() < 1e-8
            res = [abs(np.r_[x2[i], 1] @ M @ np.r_[x1[i], 1]) for i in range(5)]
            assert max(res) < 1e-10


def test_sampson_zero_on_noise_free_inliers(oracle):
    x1, x2, E = _true_E_pair(3)
    err = oracle.sampson_errors(E, x1, x2)
    assert err.max() < 1e-25
    # scale invariance and a hand value
    E2 = np.zeros((3, 3))
    E2[0, 1] = -1.0
    E2[1, 0] = 1.0   # x2^T E x1 = x2.y*x1.x - x2.x*x1.y
    e = oracle.sampson_errors(E2, np.array([[1.0, 0.0]]), np.array([[0.0, 2.0]]))
    # s = 2 ; Ex1 = (0,1,0) ; E^T x2 = (2,0,0) ; err = 4 / (0+1+4+0)
    assert e[0] == np.float32(0.8)


def test_seven_point_contains_true_F(oracle):
    x1, x2, E = _true_E_pair(5)
    K = synth.REF_K
    p1 = (np.c_[x1, np.ones(len(x1))] @ K.T)[:, :2].astype(np.float32)
    p2 = (np.c_[x2, np.ones(len(x2))] @ K.T)[:, :2].astype(np.float32)
    Fs = oracle.seven_point(p1[:7], p2[:7])
    assert 1 <= len(Fs) <= 3
    errs = [oracle.symepi_errors(F, p1, p2).max() for F in Fs]
    assert min(errs) < 1e-2   # float32 pixel rounding only
    for F in Fs:
        assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-9


def test_ransac_adaptive_matches_geometry(oracle):
    p1, p2, is_in, E = synth.two_view(1500, 11, inlier_frac=0.85, noise_px=0.3)
    r = oracle.ransac(p1, p2, synth.REF_K, prob=0.99, threshold=1.0, want_counts=True)
    assert r["ok"] == 1
    assert r["iters"] < 100
    assert r["n_inliers"] == int(r["mask"].sum())
    assert (r["mask"].astype(bool) & ~is_in).sum() <= 5
    assert r["n_inliers"] > 0.9 * is_in.sum()
    M = r["model"]
    assert min(np.abs(M - E).max(), np.abs(M + E).max()) < 0.05


def test_ransac_degenerate_counts(oracle):
    p1, p2, _, _ = synth.two_view(5, 1, inlier_frac=1.0, noise_px=0.0)
    sentinel = np.full(4, 7, np.uint8)
    r = oracle.ransac(p1[:4], p2[:4], synth.REF_K, mask_init=sentinel)
    assert r["ok"] == 0 and r["mask"].tolist() == [7, 7, 7, 7]   # n < 5: mask untouched
    r = oracle.ransac(p1, p2, synth.REF_K)
    assert r["ok"] == 1 and r["mask"].tolist() == [1] * 5        # n == 5: direct solve


def test_ransac_fixed_count_is_replayable(oracle):
    # the adaptive loop is a prefix replay of the fixed-count scores
    p1, p2, _, _ = synth.two_view(800, 21, inlier_frac=0.8)
    ra = oracle.ransac(p1, p2, synth.REF_K, want_counts=True)
    table = oracle.sample_table(800, 5, ra["iters"])     # the rows the adaptive loop drew
    rf = oracle.ransac(p1, p2, synth.REF_K, fixed_count=ra["iters"], sample_table=table,
                       want_counts=True)
    assert (ra["counts"][:ra["iters"]] == rf["counts"][:ra["iters"]]).all()
    assert (ra["mask"] == rf["mask"]).all()


# ---- pose recovery (main.cpp:420) -----------------------------------------

def test_svd3_factorises(oracle):
    rng = np.random.default_rng(5)
    for _ in range(200):
        A = rng.normal(size=(3, 3))
        ok, U, S, V = oracle.svd3(A)
        assert ok == 1
        assert np.abs(U @ np.diag(S) @ V.T - A).max() < 1e-13
        assert np.abs(np.abs(S) - np.linalg.svd(A)[1]).max() < 1e-13
        assert abs(np.linalg.det(U) - 1) < 1e-13 and np.abs(V.T @ V - np.eye(3)).max() < 1e-13
    assert oracle.svd3(np.zeros((3, 3)))[0] == 0
    assert oracle.svd3(np.outer([1.0, 2, 3], [4.0, 5, 6]))[0] == 0  # rank 1


def test_decompose_essential_candidates(oracle):
    from sfm_amd import synth
    for seed in range(10):
        _, _, _, E, R, t = synth.two_view(50, seed, with_pose=True)
        ok, R1, R2, tt = oracle.decompose_essential(E)
        assert ok == 1
        for Rc in (R1, R2):
            assert np.abs(Rc @ Rc.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Rc) - 1) < 1e-12
            tx = np.array([[0, -tt[2], tt[1]], [tt[2], 0, -tt[0]], [-tt[1], tt[0], 0]])
            Ec = tx @ Rc
            Ec /= np.linalg.norm(Ec)
            assert min(np.abs(Ec - E).max(), np.abs(Ec + E).max()) < 1e-12
        assert abs(np.linalg.norm(tt) - 1) < 1e-12
        assert min(np.abs(R1 - R).max(), np.abs(R2 - R).max()) < 1e-9
        assert min(np.abs(tt - t).max(), np.abs(tt + t).max()) < 1e-9


def test_recover_pose_finds_the_planted_pose(oracle):
    from sfm_amd import synth
    for seed in range(12):
        p1, p2, is_in, E, R, t = synth.two_view(700, seed, inlier_frac=0.8, with_pose=True)
        r = oracle.recover_pose(E, p1, p2, synth.REF_K)
        assert np.abs(r["R"] - R).max() < 1e-9 and np.abs(r["t"] - t).max() < 1e-9
        assert r["n_good"] == r["votes"].max() == int(r["mask"].sum())
        assert r["mask"][is_in].mean() > 0.99
        # E is defined up to sign; the pose is not
        r2 = oracle.recover_pose(-E, p1, p2, synth.REF_K)
        assert np.abs(r2["R"] - R).max() < 1e-9 and np.abs(r2["t"] - t).max() < 1e-9
        assert (r2["mask"] == r["mask"]).all()
        # swapping the images gives the inverse pose
        r3 = oracle.recover_pose(E.T, p2, p1, synth.REF_K)
        assert np.abs(r3["R"] - R.T).max() < 1e-9
        tb = -R.T @ t
        assert np.abs(r3["t"] - tb / np.linalg.norm(tb)).max() < 1e-9


def test_recover_pose_mask_and_distance(oracle):
    from sfm_amd import synth
    p1, p2, is_in, E, R, t = synth.two_view(500, 3, inlier_frac=0.75, with_pose=True)
    K = synth.REF_K
    full = oracle.recover_pose(E, p1, p2, K)
    # rows masked out do not vote and stay 0
    m = (np.arange(500) % 3 != 0).astype(np.uint8)
    part = oracle.recover_pose(E, p1, p2, K, mask=m)
    assert (part["mask"] == (full["mask"] & m)).all() and part["n_good"] == int((full["mask"] & m).sum())
    # synthetic depths are 17..37.4 baselines: a threshold below them rejects everything planted
    near = oracle.recover_pose(E, p1, p2, K, distance_thresh=10.0)
    assert near["mask"][is_in].sum() == 0
    mid = oracle.recover_pose(E, p1, p2, K, distance_thresh=27.0)
    assert 0 < mid["mask"][is_in].sum() < full["mask"][is_in].sum()
    # no points: OpenCV's tie order picks the first candidate
    ok, R1, _, tt = oracle.decompose_essential(E)
    none = oracle.recover_pose(E, np.zeros((0, 2)), np.zeros((0, 2)), K)
    assert none["n_good"] == 0 and (none["R"] == R1).all() and (none["t"] == tt).all()
    assert oracle.recover_pose(np.zeros((3, 3)), p1, p2, K)["n_good"] == -1


# ---- three-view track join (main.cpp:442-456) -----------------------------

def _dm(q, t):
    m = np.zeros(len(q), dtype=pyoracle_dtype())
    m["queryIdx"] = q
    m["trainIdx"] = t
    return m


def pyoracle_dtype():
    from oracle import pyoracle
    return pyoracle.DMATCH_DTYPE


def test_track_join_is_the_nested_loop(oracle):
    ab = _dm([10, 11, 12, 13], [5, 3, 9, 3])
    bc = _dm([3, 3, 5, 7, 9], [100, 101, 102, 103, 104])
    assert oracle.track_join(ab, bc).tolist() == [[10, 5, 102], [11, 3, 100], [11, 3, 101], [12, 9, 104],
                                                  [13, 3, 100], [13, 3, 101]]
    assert oracle.track_join(ab[:0], bc).shape == (0, 3) and oracle.track_join(ab, bc[:0]).shape == (0, 3)
    # against a literal Python statement of the loop on random (unsorted, repeating) lists
    rng = np.random.default_rng(4)
    a = _dm(rng.integers(0, 50, 80), rng.integers(0, 40, 80))
    b = _dm(rng.integers(0, 40, 70), rng.integers(0, 60, 70))
    want = [[int(x["queryIdx"]), int(x["trainIdx"]), int(y["trainIdx"])] for x in a for y in b
            if x["trainIdx"] == y["queryIdx"]]
    assert oracle.track_join(a, b).tolist() == want


# ---- resection: EPnP + its RANSAC loop (bundle_adjustment.h:139-155) ----------------

def _pnp_scene(n, noise, out_frac, seed):
    from sfm_amd import synth
    rng = np.random.default_rng(seed)
    K = synth.REF_K
    R = synth._rot(rng.normal(size=3), np.deg2rad(rng.uniform(5, 40)))
    t = np.array([rng.normal() * 30, rng.normal() * 30, 600 + rng.normal() * 50])
    X = rng.uniform(-150, 150, (n, 3))
    Xc = X @ R.T + t
    u = Xc @ K.T
    u = u[:, :2] / u[:, 2:] + rng.normal(0, noise, (n, 2))
    no = int(n * out_frac)
    u[:no] = rng.uniform(0, 1500, (no, 2))
    return K, X, u, R, t


@pytest.mark.parametrize("n", [5, 6, 10, 100, 2000])
def test_epnp_recovers_a_planted_pose_exactly_on_noise_free_points(oracle, n):
    K, X, u, R, t = _pnp_scene(n, 0.0, 0.0, 100 + n)
    r = oracle.epnp(X, u, K)
    assert r["ok"] == 1
    assert np.abs(r["R"] - R).max() < 1e-9 and np.abs(r["t"] - t).max() < 1e-6
    assert abs(np.linalg.det(r["R"]) - 1.0) < 1e-12 and np.abs(r["R"] @ r["R"].T - np.eye(3)).max() < 1e-12


def test_pnp_ransac_loop_semantics(oracle):
    K, X, u, R, t = _pnp_scene(2000, 0.5, 0.3, 7)
    o = oracle.pnp_ransac(X, u, K)
    assert o["ok"] == 1 and 1350 <= o["n_inliers"] <= 1400 and (o["mask"][:600] == 0).mean() > 0.99
    assert np.abs(o["R"] - R).max() < 1e-3 and np.abs(o["tvec"] - t).max() < 1.0
    # the iteration count follows RANSACUpdateNumIters: 30 % outliers, 5-point samples,
    # confidence 0.99 -> ln(0.01) / ln(1 - 0.7^5) = 25
    assert o["iters"] == oracle.update_num_iters(0.99, 1 - o["n_inliers"] / 2000.0, 5, 100) == 25
    # every reported inlier reprojects within the threshold, evaluated like the loop does (float32)
    Xf = X.astype(np.float32).astype(np.float64)
    p = Xf @ o["R"].T + o["tvec"]
    # (the mask belongs to the winning hypothesis, the pose is the refit over its inliers: close, not equal)
    pu = (p[:, :2] / p[:, 2:]) * [K[0, 0], K[1, 1]] + [K[0, 2], K[1, 2]]
    d2 = ((pu - u) ** 2).sum(1)
    assert (d2[o["mask"] == 1] < 64.0 * 1.5).all() and (d2[o["mask"] == 0] > 64.0 / 1.5).all()
    # rotation vector: Rodrigues' formula gives the matrix back
    th = np.linalg.norm(o["rvec"])
    k = o["rvec"] / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    assert np.abs(np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx - o["R"]).max() < 1e-12
    # a single iteration, and too few points
    one = oracle.pnp_ransac(X, u, K, iterations_count=1)
    assert one["iters"] <= 1
    assert oracle.pnp_ransac(X[:4], u[:4], K)["ok"] == -1


def test_rotation_vector_special_angles(oracle):
    assert np.abs(oracle.rotation_vector(np.eye(3))).max() == 0.0
    Rpi = np.diag([1.0, -1.0, -1.0])                       # half turn about x
    assert np.allclose(oracle.rotation_vector(Rpi), [np.pi, 0, 0])
    a = 0.3
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    assert np.allclose(oracle.rotation_vector(Rz), [0, 0, a])


@pytest.mark.parametrize("kind", ["sift", "uniform", "ties", "collisions", "ragged"])
def test_integer_knn_variant_returns_the_same_bytes(oracle, kind):
    """The variant the benchmark times as the CPU baseline (bytes + int32 sums,
    four query rows per pass) against the float32 restatement: every record and
    count equal, including tie order, distances at and above 2^22 where distinct
    integers share a float root, and a row count that is no multiple of four."""
    from sfm_amd import synth
    rng = np.random.default_rng(99)
    if kind == "sift":
        Q, T = synth.sift_like(203, 5), synth.sift_like(517, 6)
    elif kind == "uniform":
        Q = rng.integers(0, 256, (131, 128)).astype(np.float32)
        T = rng.integers(0, 256, (260, 128)).astype(np.float32)
    elif kind == "ties":
        Q = (rng.integers(0, 2, (97, 128)) * 50).astype(np.float32)
        T = (rng.integers(0, 2, (301, 128)) * 50).astype(np.float32)
        T[40:80] = T[0:40]
    elif kind == "collisions":
        # far-apart rows: every distance squared is above 2^22
        Q = rng.integers(0, 30, (64, 128)).astype(np.float32)
        T = rng.integers(215, 256, (333, 128)).astype(np.float32)
    else:
        Q, T = synth.sift_like(6, 7), synth.sift_like(3, 8)
    for k in (1, 2):
        for nt in (len(T), 1, 0):
            a, ca = oracle.knn_match(Q, T[:nt], k, nthreads=3)
            b, cb = oracle.knn_match(Q, T[:nt], k, nthreads=3, u8path=True)
            assert a.tobytes() == b.tobytes() and (ca == cb).all()
    if kind == "collisions":
        d2 = ((Q[:, None, :] - T[None, :, :]) ** 2).sum(-1)
        assert d2.min() >= 2 ** 22
    # non-byte values fall through to the float path
    Q2 = Q.copy()
    Q2[0, 0] = 0.5
    a, ca = oracle.knn_match(Q2, T, 2)
    b, cb = oracle.knn_match(Q2, T, 2, u8path=True)
    assert a.tobytes() == b.tobytes()


def test_oracle_runs_clean_under_address_and_undefined_sanitizers():
    """`make -C oracle asan`: the oracle's sources built with
    -fsanitize=address,undefined and run through every entry point on small
    known cases (oracle/selfcheck.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "--no-print-directory", "asan"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "oracle selfcheck ok" in p.stdout
