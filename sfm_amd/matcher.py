"""Host-side mirror of the reference's interface for the matching path.

The reference calls, per image pair (/root/reference/main.cpp:181-231):

    matcher->knnMatch(descriptorsL, descriptorsR, matches, 2);       // :184
    if (m[0].distance < 0.7 * m[1].distance) ...                      // :192
    findEssentialMat(ptsL, ptsR, K, FM_RANSAC, 0.99, 1.0, inliers);   // :224

The classes below keep those names and argument meanings and forward to the
C ABI of libsfm_gpu.so.  numpy arrays stand in for cv::Mat; a structured array
with the cv::DMatch fields stands in for vector<vector<DMatch>>.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (DMATCH_DTYPE, DESC_F32, DESC_U8, MODEL_ESSENTIAL, MODEL_FUNDAMENTAL,
                   RANSAC_ADAPTIVE, RANSAC_FIXED, MatchParams, SfmGpuError, ptr)

FM_RANSAC = 8  # cv::FM_RANSAC == cv::RANSAC


class Context:
    """One GPU + one HIP stream (sfm_gpu_create)."""

    def __init__(self, device_id=0, stream=None, timing_build=False):
        # timing_build: the library variant that carries the kernels' timing
        # ablations (measurement scripts only)
        self._L = _lib.timing_lib() if timing_build else _lib.lib()
        h = ctypes.c_void_p()
        rc = self._L.sfm_gpu_create(device_id, ctypes.c_void_p(stream) if stream else None,
                                    ctypes.byref(h))
        if rc != 0:
            raise SfmGpuError(rc, "sfm_gpu_create failed (no GPU, or bad device id %d)" % device_id)
        self.handle = h
        self.device_id = device_id
        self.stream = stream  # hipStream_t the context launches on, None: a stream of its own

    def check(self, rc):
        if rc != 0:
            raise SfmGpuError(rc, self._L.sfm_gpu_last_error_string(self.handle).decode())

    def synchronize(self):
        self.check(self._L.sfm_gpu_synchronize(self.handle))

    def close(self):
        if self.handle:
            self._L.sfm_gpu_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # measurement support
    def profile_enable(self, on=True):
        self.check(self._L.sfm_gpu_profile_enable(self.handle, 1 if on else 0))

    def profile_select(self, kernel=None):
        self.check(self._L.sfm_gpu_profile_select(self.handle, kernel.encode() if kernel else None))

    def profile_reset(self):
        self.check(self._L.sfm_gpu_profile_reset(self.handle))

    def profile_query(self, kernel):
        ms = ctypes.c_double(0)
        n = ctypes.c_int64(0)
        self.check(self._L.sfm_gpu_profile_query(self.handle, kernel.encode(), ctypes.byref(ms),
                                                 ctypes.byref(n)))
        return ms.value, n.value


def _desc_args(a):
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("descriptors must be 2-D")
    if a.dtype == np.float32:
        dtype = DESC_F32
    elif a.dtype == np.uint8:
        dtype = DESC_U8
    else:
        raise TypeError("descriptors must be float32 (CV_32F) or uint8 (CV_8U)")
    if a.shape[0] > 0 and a.strides[1] != a.itemsize:
        a = np.ascontiguousarray(a)
    step = a.strides[0] if a.shape[0] > 0 else a.shape[1] * a.itemsize
    return a, dtype, step


class BFMatcher:
    """cv::BFMatcher(NORM_L2) stand-in; only knnMatch, the method main.cpp:184 calls."""

    def __init__(self, ctx):
        self.ctx = ctx

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2):
        """Returns (matches, counts): matches[nq, k] with cv::DMatch fields,
        counts[nq] = number of valid entries per row (= min(k, n_train))."""
        q, qd, qstep = _desc_args(queryDescriptors)
        t, td, tstep = _desc_args(trainDescriptors)
        if qd != td:
            raise TypeError("query and train descriptor types differ")
        nq, nt = q.shape[0], t.shape[0]
        dim = q.shape[1]
        if nt > 0 and t.shape[1] != dim:
            raise ValueError("query and train descriptor lengths differ")
        matches = np.zeros((nq, k), dtype=DMATCH_DTYPE)
        counts = np.zeros(nq, dtype=np.int32)
        L = self.ctx._L
        self.ctx.check(L.sfm_gpu_knn_match(self.ctx.handle, ptr(q), nq, qstep, ptr(t), nt, tstep,
                                           dim, qd, k, ptr(matches), ptr(counts)))
        return matches, counts


def ratio_test(matches, counts, ratio=0.7):
    """main.cpp:189-197: rows whose best distance < ratio * second-best (double)."""
    L = _lib.lib()
    matches = np.ascontiguousarray(matches)
    nq, k = matches.shape
    keep = np.zeros(max(nq, 1), dtype=np.int32)
    n = ctypes.c_int32(0)
    rc = L.sfm_gpu_ratio_test(ptr(matches), ptr(np.ascontiguousarray(counts, dtype=np.int32)), nq,
                              k, float(ratio), ptr(keep), ctypes.byref(n))
    if rc != 0:
        raise SfmGpuError(rc, "sfm_gpu_ratio_test: bad arguments")
    return keep[:n.value].copy()


def _ransac(ctx, pts1, pts2, K, model, prob, threshold, max_iters, mode, sample_table, n_hyp,
            mask):
    pts1 = np.ascontiguousarray(pts1, dtype=np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 2)
    n = pts1.shape[0]
    if pts2.shape[0] != n:
        raise ValueError("point arrays differ in length")
    if mask is None:
        mask = np.zeros(n, dtype=np.uint8)
    Kc = None if K is None else np.ascontiguousarray(K, dtype=np.float64)
    st = None
    if sample_table is not None:
        st = np.ascontiguousarray(sample_table, dtype=np.int32)
        n_hyp = st.shape[0]
    out = np.zeros((3, 3), dtype=np.float64)
    n_inl = ctypes.c_int32(0)
    iters = ctypes.c_int32(0)
    ok = ctypes.c_int32(0)
    ctx.check(ctx._L.sfm_gpu_ransac(ctx.handle, ptr(pts1), ptr(pts2), n, ptr(Kc), model, prob,
                                    threshold, max_iters, mode, ptr(st), n_hyp, ptr(mask), ptr(out),
                                    ctypes.byref(n_inl), ctypes.byref(iters), ctypes.byref(ok)))
    return dict(ok=ok.value, model=out, mask=mask, n_inliers=n_inl.value, iters=iters.value)


def ransac_candidates(ctx):
    """All candidate models ([k, 3, 3]) of the minimal sample that won the last
    findEssentialMat / findFundamentalMat call on ctx; with exactly 5 (7) points
    this is the stack of solutions OpenCV returns as one 3k x 3 matrix."""
    out = np.zeros((10, 3, 3), dtype=np.float64)
    n = ctypes.c_int32(0)
    ctx.check(ctx._L.sfm_gpu_ransac_candidates(ctx.handle, ptr(out), 10, ctypes.byref(n)))
    return out[:n.value].copy()


def findEssentialMat(ctx, points1, points2, cameraMatrix, method=FM_RANSAC, prob=0.999,
                     threshold=1.0, maxIters=1000, mask=None, mode=RANSAC_ADAPTIVE,
                     sample_table=None, n_hyp=0):
    """cv::findEssentialMat(points1, points2, cameraMatrix, method, prob,
    threshold[, maxIters], mask) as called at main.cpp:224.  Returns (E, mask);
    E is None when no model was found (OpenCV returns an empty Mat), and the
    mask passed in is then left untouched."""
    if method != FM_RANSAC:
        raise NotImplementedError("only FM_RANSAC is on the accelerated path")
    r = _ransac(ctx, points1, points2, cameraMatrix, MODEL_ESSENTIAL, prob, threshold, maxIters,
                mode, sample_table, n_hyp, mask)
    r["E"] = r["model"] if r["ok"] else None
    return r["E"], r["mask"], r


def findFundamentalMat(ctx, points1, points2, method=FM_RANSAC, ransacReprojThreshold=3.0,
                       confidence=0.99, maxIters=1000, mask=None, mode=RANSAC_ADAPTIVE,
                       sample_table=None, n_hyp=0):
    """cv::findFundamentalMat(points1, points2, FM_RANSAC, threshold, confidence)."""
    if method != FM_RANSAC:
        raise NotImplementedError("only FM_RANSAC is on the accelerated path")
    r = _ransac(ctx, points1, points2, None, MODEL_FUNDAMENTAL, confidence, ransacReprojThreshold,
                maxIters, mode, sample_table, n_hyp, mask)
    r["F"] = r["model"] if r["ok"] else None
    return r["F"], r["mask"], r


def recoverPose(ctx, E, points1, points2, cameraMatrix, distanceThresh=50.0, mask=None):
    """cv::recoverPose(E, points1, points2, cameraMatrix, R, t[, distanceThresh, mask])
    as called at main.cpp:420 and bundle_adjustment.h:80.  Returns
    (n_good, R, t, mask) like cv2.recoverPose: R 3x3, t 3x1 with |t| = 1, mask
    n bytes (the mask passed in restricts which rows vote)."""
    E = np.ascontiguousarray(E, dtype=np.float64).reshape(3, 3)
    pts1 = np.ascontiguousarray(points1, dtype=np.float64).reshape(-1, 2)
    pts2 = np.ascontiguousarray(points2, dtype=np.float64).reshape(-1, 2)
    n = pts1.shape[0]
    if pts2.shape[0] != n:
        raise ValueError("point arrays differ in length")
    Kc = np.ascontiguousarray(cameraMatrix, dtype=np.float64).reshape(3, 3)
    if mask is None:
        m = np.ones(n, dtype=np.uint8)
    else:
        m = np.array(mask, dtype=np.uint8).reshape(-1)
        if m.shape[0] != n:
            raise ValueError("mask and points differ in length")
    R = np.zeros((3, 3), dtype=np.float64)
    t = np.zeros((3, 1), dtype=np.float64)
    good = ctypes.c_int32(0)
    ctx.check(ctx._L.sfm_gpu_recover_pose(ctx.handle, ptr(E), ptr(pts1), ptr(pts2), n, ptr(Kc),
                                          float(distanceThresh), ptr(R), ptr(t), ptr(m),
                                          ctypes.byref(good)))
    return good.value, R, t, m


def solvePnPRansac(ctx, objectPoints, imagePoints, cameraMatrix, iterationsCount=100, reprojectionError=8.0,
                   confidence=0.99):
    """cv::solvePnPRansac(objectPoints, imagePoints, cameraMatrix, zeros, rvec, tvec,
    false, iterationsCount, reprojectionError, confidence, inliers, SOLVEPNP_EPNP) as
    Resection() calls it (bundle_adjustment.h:154).  Returns (ok, rvec, tvec, inliers,
    info) with inliers = indices of the inlier points like OpenCV's output array and
    info = dict(R, mask, n_inliers, iters)."""
    X = np.ascontiguousarray(objectPoints, dtype=np.float64).reshape(-1, 3)
    u = np.ascontiguousarray(imagePoints, dtype=np.float64).reshape(-1, 2)
    if X.shape[0] != u.shape[0]:
        raise ValueError("object and image points differ in length")
    n = X.shape[0]
    Kc = np.ascontiguousarray(cameraMatrix, dtype=np.float64).reshape(3, 3)
    rvec = np.zeros(3); tvec = np.zeros(3); R = np.zeros((3, 3))
    mask = np.zeros(max(n, 1), np.uint8)
    ni = ctypes.c_int32(0); it = ctypes.c_int32(0); ok = ctypes.c_int32(0)
    ctx.check(ctx._L.sfm_gpu_pnp_ransac(ctx.handle, ptr(X), ptr(u), n, ptr(Kc), int(iterationsCount),
                                        float(reprojectionError), float(confidence), ptr(rvec), ptr(tvec), ptr(R),
                                        ptr(mask), ctypes.byref(ni), ctypes.byref(it), ctypes.byref(ok)))
    mask = mask[:n]
    return bool(ok.value), rvec, tvec, np.flatnonzero(mask).astype(np.int32), dict(
        R=R, mask=mask, n_inliers=ni.value, iters=it.value)


def track_join(ctx, matches_ab, matches_bc, capacity=None):
    """The three-view join of main.cpp:442-456: [n, 3] int32 keypoint indices
    (image i, image i+1, image j), in the reference's loop order.  matches_bc
    must be in ascending queryIdx order (match() emits it so)."""
    ab = np.ascontiguousarray(matches_ab, dtype=DMATCH_DTYPE).reshape(-1)
    bc = np.ascontiguousarray(matches_bc, dtype=DMATCH_DTYPE).reshape(-1)
    if capacity is None:
        capacity = max(len(ab), 1)
    out = np.zeros((capacity, 3), dtype=np.int32)
    n = ctypes.c_int32(0)
    rc = ctx._L.sfm_gpu_track_join(ctx.handle, ptr(ab), len(ab), ptr(bc), len(bc), ptr(out), capacity,
                                   ctypes.byref(n))
    if rc == _lib.ERR_CAPACITY and capacity < n.value:
        return track_join(ctx, ab, bc, capacity=n.value)   # duplicates in bc: more triples than matches
    ctx.check(rc)
    return out[:n.value].copy()


def pose_votes(ctx):
    """Vote counts of the four candidates in the last recoverPose call."""
    v = np.zeros(4, dtype=np.int32)
    ctx.check(ctx._L.sfm_gpu_debug_pose_votes(ctx.handle, ptr(v)))
    return v


class Image:
    """One image's descriptors (+ keypoints) resident on the GPU
    (descriptors[i] / keypoints[i] of main.cpp:175-178)."""

    def __init__(self, ctx, descriptors, keypoints_xy=None, device_ptrs=None):
        """descriptors: numpy [n,128] float32/uint8; keypoints_xy: numpy [n,2]
        float32.  device_ptrs=(desc_ptr, n, step, dtype, kp_ptr) uses memory
        already on the GPU instead."""
        self.ctx = ctx
        h = ctypes.c_void_p()
        L = ctx._L
        if device_ptrs is not None:
            dptr, n, step, dtype, kptr = device_ptrs
            rc = L.sfm_gpu_image_create(ctx.handle, ctypes.c_void_p(dptr), n, step, 128, dtype,
                                        ctypes.c_void_p(kptr) if kptr else None, 1, ctypes.byref(h))
        else:
            d, dtype, step = _desc_args(descriptors)
            kp = None
            if keypoints_xy is not None:
                kp = np.ascontiguousarray(keypoints_xy, dtype=np.float32).reshape(-1, 2)
                if kp.shape[0] != d.shape[0]:
                    raise ValueError("keypoints and descriptors differ in length")
            rc = L.sfm_gpu_image_create(ctx.handle, ptr(d), d.shape[0], step, d.shape[1], dtype,
                                        ptr(kp), 0, ctypes.byref(h))
        ctx.check(rc)
        self.handle = h
        self.n = L.sfm_gpu_image_rows(h)

    def close(self):
        if self.handle and self.ctx.handle:
            self.ctx._L.sfm_gpu_image_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def knn_match_images(ctx, query_img, train_img, k=2, out_device=None):
    """knnMatch between prepared images.  out_device=(matches_ptr, counts_ptr)
    leaves the results on the GPU and only enqueues the work."""
    L = ctx._L
    if out_device is not None:
        ctx.check(L.sfm_gpu_knn_match_images(ctx.handle, query_img.handle, train_img.handle, k,
                                             ctypes.c_void_p(out_device[0]),
                                             ctypes.c_void_p(out_device[1]), 1))
        return None
    matches = np.zeros((query_img.n, k), dtype=DMATCH_DTYPE)
    counts = np.zeros(query_img.n, dtype=np.int32)
    ctx.check(L.sfm_gpu_knn_match_images(ctx.handle, query_img.handle, train_img.handle, k,
                                         ptr(matches), ptr(counts), 0))
    return matches, counts


def _match_params(K, ratio, model, mode, prob, threshold, max_iters, n_hyp):
    p = MatchParams()
    p.ratio = ratio
    p.model = model
    p.mode = mode
    Kc = np.ascontiguousarray(K if K is not None else np.eye(3), dtype=np.float64).reshape(9)
    for i in range(9):
        p.K[i] = Kc[i]
    p.prob = prob
    p.threshold_px = threshold
    p.max_iters = max_iters
    p.n_hyp = n_hyp
    return p


class PairBatch:
    """One batch of pairs with its argument block built once, for hosts that
    keep the GPU fed: begin() puts the batch on the stream and returns (fixed
    mode), end() waits for the oldest batch begun on the context and returns
    its (n_good, n_inliers).  At most two batches may be in flight."""

    def __init__(self, ctx, images, pairs, K, ratio=0.7, model=MODEL_ESSENTIAL, mode=RANSAC_ADAPTIVE,
                 prob=0.99, threshold=1.0, max_iters=1000, n_hyp=0, records_dev=None, match_capacity=0,
                 records_host=None):
        """records_dev: device address of n_pairs * match_capacity cv::DMatch records
        that receive each pair's selectMatches (main.cpp:225-231, 267); records_host:
        address of pinned host memory of the same shape, filled behind the batch."""
        self.ctx = ctx
        self._images = list(images)             # keeps the handles alive
        self._pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        self.n_pairs = self._pairs.shape[0]
        self._arr = (ctypes.c_void_p * len(images))(*[im.handle for im in images])
        self._p = _match_params(K, ratio, model, mode, prob, threshold, max_iters, n_hyp)
        self._args = (ctx.handle, self._arr, len(images), ptr(self._pairs), self.n_pairs, ctypes.byref(self._p))
        self._records = records_dev is not None
        if self._records:
            self._args = self._args + (ctypes.c_void_p(records_dev), int(match_capacity),
                                       ctypes.c_void_p(records_host) if records_host else None)

    def begin(self):
        if self._records:
            self.ctx.check(self.ctx._L.sfm_gpu_match_pairs_begin_records(*self._args))
        else:
            self.ctx.check(self.ctx._L.sfm_gpu_match_pairs_begin(*self._args))

    def end(self):
        n_good = np.zeros(self.n_pairs, dtype=np.int32)
        n_inl = np.zeros(self.n_pairs, dtype=np.int32)
        self.ctx.check(self.ctx._L.sfm_gpu_match_pairs_end(self.ctx.handle, ptr(n_good), ptr(n_inl), self.n_pairs))
        return n_good, n_inl


def match_pairs(ctx, images, pairs, K, ratio=0.7, model=MODEL_ESSENTIAL, mode=RANSAC_ADAPTIVE,
                prob=0.99, threshold=1.0, max_iters=1000, n_hyp=0, want_matches=False):
    """The loop body of match() (main.cpp:175-231, 267) over a list of pairs.
    Returns (n_good, n_inliers[, inlier_matches list])."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    n_pairs = pairs.shape[0]
    arr = (ctypes.c_void_p * len(images))(*[im.handle for im in images])
    p = _match_params(K, ratio, model, mode, prob, threshold, max_iters, n_hyp)
    n_good = np.zeros(n_pairs, dtype=np.int32)
    n_inl = np.zeros(n_pairs, dtype=np.int32)
    cap = 0
    out = None
    if want_matches:
        cap = max([images[i].n for i in pairs[:, 0]] + [1])
        out = np.zeros((n_pairs, cap), dtype=DMATCH_DTYPE)
    ctx.check(ctx._L.sfm_gpu_match_pairs(ctx.handle, arr, len(images), ptr(pairs), n_pairs, ctypes.byref(p),
                                         ptr(n_good), ptr(n_inl), ptr(out), cap))
    if want_matches:
        return n_good, n_inl, [out[i, :n_inl[i]].copy() for i in range(n_pairs)]
    return n_good, n_inl


def match_pairs_pose(ctx, images, pairs, K, ratio=0.7, mode=RANSAC_ADAPTIVE, prob=0.99, threshold=1.0,
                     max_iters=1000, n_hyp=0, distanceThresh=50.0, want_matches=False):
    """match_pairs (essential model) followed per pair by main.cpp:400-420: a
    second findEssentialMat (prob 0.999, 1 px, 1000 iterations) over the pair's
    inlier matches and recoverPose with that matrix on all of them (the
    reference does this for the first pair).  Returns a dict: n_good, n_inliers,
    E [n,3,3] (of the second RANSAC), R [n,3,3], t [n,3], n_pose_good [n]
    (-1: no model), and inlier_matches when asked for."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    n_pairs = pairs.shape[0]
    arr = (ctypes.c_void_p * len(images))(*[im.handle for im in images])
    p = _match_params(K, ratio, MODEL_ESSENTIAL, mode, prob, threshold, max_iters, n_hyp)
    n_good = np.zeros(n_pairs, dtype=np.int32)
    n_inl = np.zeros(n_pairs, dtype=np.int32)
    E = np.zeros((n_pairs, 3, 3))
    R = np.zeros((n_pairs, 3, 3))
    t = np.zeros((n_pairs, 3))
    n_pose = np.zeros(n_pairs, dtype=np.int32)
    cap = 0
    out = None
    if want_matches:
        cap = max([images[i].n for i in pairs[:, 0]] + [1])
        out = np.zeros((n_pairs, cap), dtype=DMATCH_DTYPE)
    ctx.check(ctx._L.sfm_gpu_match_pairs_pose(ctx.handle, arr, len(images), ptr(pairs), n_pairs, ctypes.byref(p),
                                              ptr(n_good), ptr(n_inl), ptr(out), cap, float(distanceThresh),
                                              ptr(E), ptr(R), ptr(t), ptr(n_pose)))
    res = dict(n_good=n_good, n_inliers=n_inl, E=E, R=R, t=t, n_pose_good=n_pose)
    if want_matches:
        res["inlier_matches"] = [out[i, :n_inl[i]].copy() for i in range(n_pairs)]
    return res


def track_join_sequence(ctx, match_lists, capacity=None):
    """The joins of main.cpp:432-456 for a whole sequence in one call: list l
    against list l+1.  Returns a list of len(match_lists)-1 arrays [n, 3]."""
    lists = [np.ascontiguousarray(m, dtype=DMATCH_DTYPE).reshape(-1) for m in match_lists]
    n_lists = len(lists)
    sizes = np.array([len(m) for m in lists], dtype=np.int32)
    mcap = int(max([1] + [len(m) for m in lists]))
    packed = np.zeros((max(n_lists, 1), mcap), dtype=DMATCH_DTYPE)
    for l, m in enumerate(lists):
        packed[l, :len(m)] = m
    if capacity is None:
        capacity = int(max(1, sizes[:-1].sum())) if n_lists > 1 else 1
    out = np.zeros((capacity, 3), dtype=np.int32)
    offs = np.zeros(max(n_lists, 1), dtype=np.int32)
    rc = ctx._L.sfm_gpu_track_join_sequence(ctx.handle, ptr(packed), ptr(sizes), mcap, n_lists, ptr(out), capacity,
                                            ptr(offs))
    if rc == _lib.ERR_CAPACITY and n_lists > 1 and capacity < offs[n_lists - 1]:
        return track_join_sequence(ctx, lists, capacity=int(offs[n_lists - 1]))
    ctx.check(rc)
    return [out[offs[l]:offs[l + 1]].copy() for l in range(n_lists - 1)]


# ---- images in caller-owned device memory ---------------------------------

def image_packed_bytes(n):
    """Size in bytes of the device block that holds a prepared image of n rows."""
    return int(_lib.lib().sfm_gpu_image_packed_bytes(int(n)))


def prepare_many(ctx, desc_ptrs, ns, step, dtype, kp_ptrs, packed_ptrs):
    """Converts `len(ns)` images that already sit in device memory into their
    packed blocks with one launch (enqueue only).  All arguments are sequences
    of device addresses (ints) / row counts."""
    cnt = len(ns)
    d = (ctypes.c_void_p * cnt)(*desc_ptrs)
    p = (ctypes.c_void_p * cnt)(*packed_ptrs)
    k = (ctypes.c_void_p * cnt)(*kp_ptrs) if kp_ptrs is not None else None
    n = np.ascontiguousarray(ns, dtype=np.int32)
    ctx.check(ctx._L.sfm_gpu_image_prepare_many(ctx.handle, d, ptr(n), step, 128, dtype, k, p, cnt))


class PrepareBatch:
    """prepare_many with its argument arrays built once: a caller that converts
    the same device buffers again and again (a pipeline loop) then pays one
    foreign call per step instead of re-marshalling hundreds of pointers."""

    def __init__(self, ctx, desc_ptrs, ns, step, dtype, kp_ptrs, packed_ptrs):
        self.ctx = ctx
        cnt = len(ns)
        self._d = (ctypes.c_void_p * cnt)(*desc_ptrs)
        self._p = (ctypes.c_void_p * cnt)(*packed_ptrs)
        self._k = (ctypes.c_void_p * cnt)(*kp_ptrs) if kp_ptrs is not None else None
        self._n = np.ascontiguousarray(ns, dtype=np.int32)
        self._args = (ctx.handle, self._d, ptr(self._n), step, 128, dtype, self._k, self._p, cnt)

    def run(self):
        self.ctx.check(self.ctx._L.sfm_gpu_image_prepare_many(*self._args))


def stream_wait_front(waiter, source):
    """Work enqueued on `waiter` from now on starts when the batch most recently
    enqueued on `source` (another context, same GPU) is past its neighbour extraction."""
    waiter.check(waiter._L.sfm_gpu_stream_wait_front(waiter.handle, source.handle))


def check_inputs(ctx):
    """Waits for the stream and raises if a prepared image held non-u8 values."""
    ctx.check(ctx._L.sfm_gpu_check_inputs(ctx.handle))


class WrappedImage:
    """Non-owning image handle over a packed block in device memory."""

    def __init__(self, ctx, packed_ptr, n, has_keypoints=True):
        self.ctx = ctx
        h = ctypes.c_void_p()
        ctx.check(ctx._L.sfm_gpu_image_wrap(ctx.handle, ctypes.c_void_p(packed_ptr), n,
                                            1 if has_keypoints else 0, ctypes.byref(h)))
        self.handle = h
        self.n = n

    def close(self):
        if self.handle and self.ctx.handle:
            self.ctx._L.sfm_gpu_image_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
