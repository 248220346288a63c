"""ctypes bridge to the CPU oracle (libsfm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under sfm_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsfm_oracle.so")

DMATCH_DTYPE = np.dtype(
    [("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")]
)

_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        c = ctypes
        vp = c.c_void_p
        L.oracle_knn_match.argtypes = [vp, c.c_int, c.c_size_t, vp, c.c_int, c.c_size_t,
                                       c.c_int, c.c_int, vp, vp, c.c_int]
        L.oracle_knn_match.restype = c.c_int
        L.oracle_knn_match_u8path.argtypes = L.oracle_knn_match.argtypes
        L.oracle_knn_match_u8path.restype = c.c_int
        L.oracle_ratio_test.argtypes = [vp, vp, c.c_int, c.c_int, c.c_double, vp]
        L.oracle_ratio_test.restype = c.c_int
        L.oracle_rng_next.argtypes = [c.c_uint64, c.c_int, vp]
        L.oracle_rng_uniform.argtypes = [c.c_uint64, c.c_int, c.c_int, c.c_int, vp]
        L.oracle_sample_table.argtypes = [c.c_int, c.c_int, c.c_int, vp]
        L.oracle_sample_table.restype = c.c_int
        L.oracle_sample_table_f.argtypes = [c.c_int, vp, vp, c.c_int, vp]
        L.oracle_sample_table_f.restype = c.c_int
        L.oracle_sample_table_fixed.argtypes = [c.c_int, c.c_int, c.c_int, vp, vp, vp]
        L.oracle_sample_table_fixed.restype = c.c_int
        L.oracle_ransac_update_num_iters.argtypes = [c.c_double, c.c_double, c.c_int, c.c_int]
        L.oracle_ransac_update_num_iters.restype = c.c_int
        L.oracle_real_roots.argtypes = [vp, c.c_int, vp]
        L.oracle_real_roots.restype = c.c_int
        L.oracle_five_point.argtypes = [vp, vp, vp]
        L.oracle_five_point.restype = c.c_int
        L.oracle_seven_point.argtypes = [vp, vp, vp]
        L.oracle_seven_point.restype = c.c_int
        L.oracle_sampson_errors.argtypes = [vp, vp, vp, c.c_int, vp]
        L.oracle_symepi_errors.argtypes = [vp, vp, vp, c.c_int, vp]
        L.oracle_normalize_points.argtypes = [vp, c.c_int, vp, c.c_double, vp]
        L.oracle_normalize_points.restype = c.c_double
        L.oracle_ransac.argtypes = [c.c_int, vp, vp, c.c_int, vp, c.c_double, c.c_double,
                                    c.c_int, c.c_int, vp, vp, vp, vp, vp, vp]
        L.oracle_ransac.restype = c.c_int
        L.oracle_svd3.argtypes = [vp, vp, vp, vp]
        L.oracle_svd3.restype = c.c_int
        L.oracle_decompose_essential.argtypes = [vp, vp, vp, vp]
        L.oracle_decompose_essential.restype = c.c_int
        L.oracle_recover_pose.argtypes = [vp, vp, vp, c.c_int, vp, c.c_double, vp, vp, vp, vp]
        L.oracle_recover_pose.restype = c.c_int
        L.oracle_track_join.argtypes = [vp, c.c_int, vp, c.c_int, vp, c.c_int]
        L.oracle_track_join.restype = c.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def knn_match(query, train, k=2, nthreads=1, u8path=False):
    """query, train: 2-D float32 arrays (row strides honoured).  Returns
    (matches[nq,k] structured, counts[nq]).  u8path: the integer variant for
    byte-valued descriptors (same bytes out; the one bench.py times)."""
    query = np.asarray(query)
    train = np.asarray(train)
    assert query.dtype == np.float32 and train.dtype == np.float32
    nq, dim = query.shape
    nt = train.shape[0]
    assert train.shape[1] == dim or nt == 0
    assert query.strides[1] == 4 and (nt == 0 or train.strides[1] == 4)
    out = np.zeros((nq, k), dtype=DMATCH_DTYPE)
    counts = np.zeros(nq, dtype=np.int32)
    fn = lib().oracle_knn_match_u8path if u8path else lib().oracle_knn_match
    rc = fn(_p(query), nq, query.strides[0] if nq else 4 * dim,
                                _p(train), nt, train.strides[0] if nt else 4 * dim,
                                dim, k, _p(out), _p(counts), nthreads)
    assert rc == 0
    return out, counts


def ratio_test(matches, counts, ratio=0.7):
    nq, k = matches.shape
    keep = np.zeros(nq, dtype=np.int32)
    n = lib().oracle_ratio_test(_p(np.ascontiguousarray(matches)), _p(counts), nq, k, ratio, _p(keep))
    return keep[:n].copy()


def rng_next(n, seed=0xFFFFFFFFFFFFFFFF):
    out = np.zeros(n, dtype=np.uint32)
    lib().oracle_rng_next(seed, n, _p(out))
    return out


def rng_uniform(a, b, n, seed=0xFFFFFFFFFFFFFFFF):
    out = np.zeros(n, dtype=np.int32)
    lib().oracle_rng_uniform(seed, a, b, n, _p(out))
    return out


def sample_table(count, m, n_iters):
    out = np.zeros((n_iters, m), dtype=np.int32)
    rc = lib().oracle_sample_table(count, m, n_iters, _p(out))
    assert rc == 0
    return out


def sample_table_f(pts1, pts2, n_iters):
    pts1 = np.ascontiguousarray(pts1, dtype=np.float32)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float32)
    out = np.zeros((n_iters, 7), dtype=np.int32)
    got = lib().oracle_sample_table_f(pts1.shape[0], _p(pts1), _p(pts2), n_iters, _p(out))
    return out[:max(got, 0)]


def sample_table_fixed(count, m, n_rows, pts1=None, pts2=None):
    out = np.zeros((n_rows, m), dtype=np.int32)
    if pts1 is not None:
        pts1 = np.ascontiguousarray(pts1, dtype=np.float32)
        pts2 = np.ascontiguousarray(pts2, dtype=np.float32)
    rc = lib().oracle_sample_table_fixed(count, m, n_rows, _p(pts1), _p(pts2), _p(out))
    assert rc == 0
    return out


def update_num_iters(p, ep, m, max_iters):
    return lib().oracle_ransac_update_num_iters(p, ep, m, max_iters)


def real_roots(coef):
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    roots = np.zeros(10, dtype=np.float64)
    n = lib().oracle_real_roots(_p(coef), len(coef) - 1, _p(roots))
    assert n >= 0
    return roots[:n].copy()


def five_point(x1, x2):
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    x2 = np.ascontiguousarray(x2, dtype=np.float64)
    models = np.zeros((10, 3, 3), dtype=np.float64)
    n = lib().oracle_five_point(_p(x1), _p(x2), _p(models))
    return models[:n].copy()


def seven_point(x1, x2):
    x1 = np.ascontiguousarray(x1, dtype=np.float32)
    x2 = np.ascontiguousarray(x2, dtype=np.float32)
    models = np.zeros((3, 3, 3), dtype=np.float64)
    n = lib().oracle_seven_point(_p(x1), _p(x2), _p(models))
    return models[:n].copy()


def sampson_errors(E, x1, x2):
    E = np.ascontiguousarray(E, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    x2 = np.ascontiguousarray(x2, dtype=np.float64)
    err = np.zeros(x1.shape[0], dtype=np.float32)
    lib().oracle_sampson_errors(_p(E), _p(x1), _p(x2), x1.shape[0], _p(err))
    return err


def symepi_errors(F, x1, x2):
    F = np.ascontiguousarray(F, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float32)
    x2 = np.ascontiguousarray(x2, dtype=np.float32)
    err = np.zeros(x1.shape[0], dtype=np.float32)
    lib().oracle_symepi_errors(_p(F), _p(x1), _p(x2), x1.shape[0], _p(err))
    return err


def normalize_points(pts_px, K, threshold_px):
    pts_px = np.ascontiguousarray(pts_px, dtype=np.float64)
    K = np.ascontiguousarray(K, dtype=np.float64)
    out = np.zeros_like(pts_px)
    thr = lib().oracle_normalize_points(_p(pts_px), pts_px.shape[0], _p(K), threshold_px, _p(out))
    return out, thr


MODEL_ESSENTIAL = 0
MODEL_FUNDAMENTAL = 1


def ransac(pts1, pts2, K=None, model=MODEL_ESSENTIAL, prob=0.99, threshold=1.0, max_iters=1000,
           fixed_count=0, sample_table=None, mask_init=None, want_counts=False):
    """Returns dict(ok, mask, model, n_inliers, iters, counts)."""
    pts1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
    n = pts1.shape[0]
    Kc = np.ascontiguousarray(K if K is not None else np.eye(3), dtype=np.float64)
    mask = np.zeros(n, dtype=np.uint8) if mask_init is None else np.array(mask_init, dtype=np.uint8)
    model_out = np.zeros((3, 3), dtype=np.float64)
    n_inl = ctypes.c_int32(0)
    iters = ctypes.c_int32(0)
    nh = fixed_count if fixed_count > 0 else max(max_iters, 1)
    counts = np.full((nh, 10), -2, dtype=np.int32) if want_counts else None
    st = None
    if sample_table is not None:
        st = np.ascontiguousarray(sample_table, dtype=np.int32)
    ok = lib().oracle_ransac(model, _p(pts1), _p(pts2), n, _p(Kc), prob, threshold, max_iters,
                             fixed_count, _p(st), _p(mask), _p(model_out),
                             ctypes.byref(n_inl), ctypes.byref(iters), _p(counts))
    return dict(ok=ok, mask=mask, model=model_out, n_inliers=n_inl.value, iters=iters.value,
                counts=counts)


def ransac_fixed_mt(pts1, pts2, K=None, model=MODEL_ESSENTIAL, threshold=1.0, fixed_count=2048, nthreads=8):
    """Fixed-count RANSAC with the hypotheses spread over threads; same result as
    ransac(..., fixed_count=...).  Returns dict(ok, mask, model, n_inliers)."""
    pts1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
    n = pts1.shape[0]
    Kc = np.ascontiguousarray(K if K is not None else np.eye(3), dtype=np.float64)
    mask = np.zeros(n, dtype=np.uint8)
    model_out = np.zeros((3, 3), dtype=np.float64)
    n_inl = ctypes.c_int32(0)
    L = lib()
    L.oracle_ransac_fixed_mt.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]
    L.oracle_ransac_fixed_mt.restype = ctypes.c_int
    ok = L.oracle_ransac_fixed_mt(model, _p(pts1), _p(pts2), n, _p(Kc), threshold, fixed_count, nthreads, _p(mask),
                                  _p(model_out), ctypes.byref(n_inl))
    return dict(ok=ok, mask=mask, model=model_out, n_inliers=n_inl.value)


def ransac_direct_models():
    """Candidates of the calling thread's last n == m ransac() call, [k, 3, 3]."""
    out = np.zeros((10, 3, 3), dtype=np.float64)
    L = lib()
    L.oracle_ransac_direct_models.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.oracle_ransac_direct_models.restype = ctypes.c_int
    k = L.oracle_ransac_direct_models(_p(out), 10)
    return out[:max(k, 0)].copy()


def epnp(object_pts, image_pts, K):
    """EPnP pose: dict(ok, R, t)."""
    X = np.ascontiguousarray(object_pts, dtype=np.float64).reshape(-1, 3)
    u = np.ascontiguousarray(image_pts, dtype=np.float64).reshape(-1, 2)
    Kc = np.ascontiguousarray(K, dtype=np.float64)
    R = np.zeros((3, 3)); t = np.zeros(3)
    L = lib()
    L.oracle_epnp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_void_p]
    L.oracle_epnp.restype = ctypes.c_int
    ok = L.oracle_epnp(_p(X), _p(u), X.shape[0], _p(Kc), _p(R), _p(t))
    return dict(ok=ok, R=R, t=t)


def rotation_vector(R):
    R = np.ascontiguousarray(R, dtype=np.float64)
    r = np.zeros(3)
    L = lib()
    L.oracle_rotation_vector.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.oracle_rotation_vector(_p(R), _p(r))
    return r


def pnp_ransac(object_pts, image_pts, K, iterations_count=100, reprojection_error=8.0, confidence=0.99):
    """cv::solvePnPRansac(..., SOLVEPNP_EPNP) restated: dict(ok, rvec, tvec, R, mask, n_inliers, iters)."""
    X = np.ascontiguousarray(object_pts, dtype=np.float64).reshape(-1, 3)
    u = np.ascontiguousarray(image_pts, dtype=np.float64).reshape(-1, 2)
    n = X.shape[0]
    Kc = np.ascontiguousarray(K, dtype=np.float64)
    rvec = np.zeros(3); tvec = np.zeros(3); R = np.zeros((3, 3))
    mask = np.zeros(n, np.uint8)
    ni = ctypes.c_int32(0); it = ctypes.c_int32(0)
    L = lib()
    L.oracle_pnp_ransac.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.oracle_pnp_ransac.restype = ctypes.c_int
    ok = L.oracle_pnp_ransac(_p(X), _p(u), n, _p(Kc), iterations_count, reprojection_error, confidence, _p(rvec),
                             _p(tvec), _p(R), _p(mask), ctypes.byref(ni), ctypes.byref(it))
    return dict(ok=ok, rvec=rvec, tvec=tvec, R=R, mask=mask, n_inliers=ni.value, iters=it.value)


def svd3(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    U = np.zeros((3, 3)); S = np.zeros(3); V = np.zeros((3, 3))
    ok = lib().oracle_svd3(_p(A), _p(U), _p(S), _p(V))
    return ok, U, S, V


def decompose_essential(E):
    E = np.ascontiguousarray(E, dtype=np.float64)
    R1 = np.zeros((3, 3)); R2 = np.zeros((3, 3)); t = np.zeros(3)
    ok = lib().oracle_decompose_essential(_p(E), _p(R1), _p(R2), _p(t))
    return ok, R1, R2, t


def recover_pose(E, pts1, pts2, K, distance_thresh=50.0, mask=None):
    """Returns dict(n_good, R, t, mask, votes); n_good = -1 on failure."""
    E = np.ascontiguousarray(E, dtype=np.float64)
    pts1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
    Kc = np.ascontiguousarray(K, dtype=np.float64)
    n = pts1.shape[0]
    R = np.zeros((3, 3)); t = np.zeros(3)
    m = np.ones(n, dtype=np.uint8) if mask is None else np.array(mask, dtype=np.uint8)
    votes = np.zeros(4, dtype=np.int32)
    good = lib().oracle_recover_pose(_p(E), _p(pts1), _p(pts2), n, _p(Kc), distance_thresh,
                                     _p(R), _p(t), _p(m), _p(votes))
    return dict(n_good=good, R=R, t=t, mask=m, votes=votes)


def track_join(ab, bc):
    """ab, bc: DMATCH_DTYPE arrays.  Returns the [n, 3] int32 triples of
    main.cpp:442-456 in the reference's loop order."""
    ab = np.ascontiguousarray(ab, dtype=DMATCH_DTYPE)
    bc = np.ascontiguousarray(bc, dtype=DMATCH_DTYPE)
    n = lib().oracle_track_join(_p(ab), len(ab), _p(bc), len(bc), None, 0)
    assert n >= 0
    out = np.zeros((n, 3), dtype=np.int32)
    got = lib().oracle_track_join(_p(ab), len(ab), _p(bc), len(bc), _p(out), n)
    assert got == n
    return out
