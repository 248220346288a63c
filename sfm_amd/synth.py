"""Seeded synthetic inputs of the shapes the matching path sees.

Descriptors mimic what cv::SIFT::detectAndCompute hands to the matcher at
/root/reference/main.cpp:143: N x 128 float32 rows whose values are integers in
0..255 with an L2 norm near 512.  Two-view geometry uses the intrinsics
hard-coded at /root/reference/main.cpp:103-105 and the depth range of
/root/reference/cams_1/00000000_cam.txt:12.
"""
import numpy as np

# main.cpp:103-105
REF_K = np.array([[2892.33, 0.0, 823.205],
                  [0.0, 2883.18, 619.071],
                  [0.0, 0.0, 1.0]], dtype=np.float64)
IMG_W, IMG_H = 1600, 1200


def sift_like(n, seed):
    """n x 128 float32, integer-valued 0..255, SIFT-style normalisation."""
    rng = np.random.default_rng(seed)
    v = rng.exponential(1.0, size=(n, 128))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = np.minimum(v, 0.2)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = np.rint(v * 512.0)
    return np.clip(v, 0, 255).astype(np.float32)


def uniform_u8(n, seed, dim=128):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, dim)).astype(np.float32)


def planted_pair(n_query, n_train, seed_q, seed_t, seed_perm, frac=0.4, sigma=6.0):
    """Query/train descriptor sets where `frac` of the query rows have a noisy
    copy at a permuted position in the train set.  Returns (Q, T, planted)
    with planted[i] = train row of query i's copy, or -1."""
    Q = sift_like(n_query, seed_q)
    T = sift_like(n_train, seed_t)
    rng = np.random.default_rng(seed_perm)
    n_pl = int(min(n_query, n_train) * frac)
    q_rows = rng.permutation(n_query)[:n_pl]
    t_rows = rng.permutation(n_train)[:n_pl]
    noise = np.rint(rng.normal(0.0, sigma, size=(n_pl, 128)))
    T[t_rows] = np.clip(Q[q_rows] + noise, 0, 255).astype(np.float32)
    planted = np.full(n_query, -1, dtype=np.int64)
    planted[q_rows] = t_rows
    return Q, T, planted


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * kx + (1 - np.cos(ang)) * (kx @ kx)


def two_view(n, seed, inlier_frac=0.9, noise_px=0.5, K=REF_K, with_pose=False):
    """n correspondences in pixels (float32 precision, stored float64 like the
    Point2f -> Point2d widening at main.cpp:194-195).  Returns
    (pts1, pts2, is_inlier, E_true) with E_true for normalised coordinates;
    with_pose adds (R, t_unit) of X2 = R X1 + t, the pose recoverPose reports."""
    rng = np.random.default_rng(seed)
    n_in = int(round(n * inlier_frac))
    Kinv = np.linalg.inv(K)
    # sample pixels in image 1 and depths, back-project
    px = np.stack([rng.uniform(0, IMG_W, n_in), rng.uniform(0, IMG_H, n_in), np.ones(n_in)], 1)
    depth = rng.uniform(425.0, 935.0, n_in)
    X = (px @ Kinv.T) * depth[:, None]
    R = _rot(rng.normal(size=3), np.deg2rad(rng.uniform(1.0, 5.0)))
    t = rng.normal(size=3)
    t[2] *= 0.2
    t = t / np.linalg.norm(t) * 25.0
    X2 = X @ R.T + t
    p2 = X2 @ K.T
    p2 = p2[:, :2] / p2[:, 2:]
    p1 = px[:, :2] + rng.normal(0, noise_px, (n_in, 2))
    p2 = p2 + rng.normal(0, noise_px, (n_in, 2))
    n_out = n - n_in
    o1 = np.stack([rng.uniform(0, IMG_W, n_out), rng.uniform(0, IMG_H, n_out)], 1)
    o2 = np.stack([rng.uniform(0, IMG_W, n_out), rng.uniform(0, IMG_H, n_out)], 1)
    pts1 = np.concatenate([p1, o1], 0)
    pts2 = np.concatenate([p2, o2], 0)
    is_in = np.concatenate([np.ones(n_in, bool), np.zeros(n_out, bool)])
    perm = rng.permutation(n)
    pts1, pts2, is_in = pts1[perm], pts2[perm], is_in[perm]
    pts1 = pts1.astype(np.float32).astype(np.float64)
    pts2 = pts2.astype(np.float32).astype(np.float64)
    tn = t / np.linalg.norm(t)
    tx = np.array([[0, -tn[2], tn[1]], [tn[2], 0, -tn[0]], [-tn[1], tn[0], 0]])
    E = tx @ R
    E /= np.linalg.norm(E)
    if with_pose:
        return pts1, pts2, is_in, E, R, tn
    return pts1, pts2, is_in, E


def keypoints_for_pair(n_query, n_train, planted, seed, noise_px=0.5, K=REF_K, wrong_frac=0.0):
    """Keypoint coordinates (float32 n x 2) for both images of a planted pair:
    planted rows are projections of common 3-D points, the rest are random.
    `wrong_frac` of the planted descriptor matches keep a random train keypoint:
    matches that pass the ratio test but not the geometry (the reference's own
    log, info.txt:23-44, has 89-96 % of the ratio-test survivors as inliers)."""
    rng = np.random.default_rng(seed)
    kq = np.stack([rng.uniform(0, IMG_W, n_query), rng.uniform(0, IMG_H, n_query)], 1)
    kt = np.stack([rng.uniform(0, IMG_W, n_train), rng.uniform(0, IMG_H, n_train)], 1)
    rows = np.nonzero(planted >= 0)[0]
    Kinv = np.linalg.inv(K)
    px = np.concatenate([kq[rows], np.ones((len(rows), 1))], 1)
    depth = rng.uniform(425.0, 935.0, len(rows))
    X = (px @ Kinv.T) * depth[:, None]
    R = _rot(rng.normal(size=3), np.deg2rad(rng.uniform(1.0, 5.0)))
    t = rng.normal(size=3)
    t[2] *= 0.2
    t = t / np.linalg.norm(t) * 25.0
    p2 = (X @ R.T + t) @ K.T
    p2 = p2[:, :2] / p2[:, 2:]
    kq[rows] += rng.normal(0, noise_px, (len(rows), 2))
    p2 = p2 + rng.normal(0, noise_px, (len(rows), 2))
    if wrong_frac > 0.0:
        right = np.random.default_rng(seed + 7919).uniform(size=len(rows)) >= wrong_frac
        kt[planted[rows[right]]] = p2[right]
    else:
        kt[planted[rows]] = p2
    return kq.astype(np.float32), kt.astype(np.float32)


def image_sequence(n_img, n_kp, seed, shared=0.5, sigma=6.0, noise_px=0.5, K=REF_K):
    """A consistent multi-view sequence (the shape of the reference's data/
    set): one pool of 3-D points with one base descriptor each, cameras moving
    along a baseline with a small rotation per step; every image shows
    `shared` of its n_kp rows from the pool (noisy copies of the descriptors,
    noisy projections) and random rows otherwise.  Returns (descs, kps, ids)
    with ids[i][row] = pool index of the row or -1."""
    rng = np.random.default_rng(seed)
    n_world = int(n_kp * shared * 2)
    base = sift_like(n_world, seed + 1)
    Kinv = np.linalg.inv(K)
    px = np.stack([rng.uniform(-200, IMG_W + 200, n_world), rng.uniform(0, IMG_H, n_world), np.ones(n_world)], 1)
    X = (px @ Kinv.T) * rng.uniform(425.0, 935.0, n_world)[:, None]
    descs, kps, ids = [], [], []
    for i in range(n_img):
        R = _rot(np.array([0.1, 1.0, 0.05]), np.deg2rad(0.8 * i))
        t = np.array([-12.0 * i, 0.5 * i, 1.0 * i])
        Xc = X @ R.T + t
        p = Xc @ K.T
        p = p[:, :2] / p[:, 2:]
        vis = np.flatnonzero((p[:, 0] >= 0) & (p[:, 0] < IMG_W) & (p[:, 1] >= 0) & (p[:, 1] < IMG_H) & (Xc[:, 2] > 0))
        n_sh = min(int(n_kp * shared), len(vis))
        pick = rng.permutation(vis)[:n_sh]
        d = sift_like(n_kp, seed + 100 + i)
        k = np.stack([rng.uniform(0, IMG_W, n_kp), rng.uniform(0, IMG_H, n_kp)], 1)
        wid = np.full(n_kp, -1, dtype=np.int64)
        rows = rng.permutation(n_kp)[:n_sh]
        d[rows] = np.clip(base[pick] + np.rint(rng.normal(0.0, sigma, (n_sh, 128))), 0, 255).astype(np.float32)
        k[rows] = p[pick] + rng.normal(0.0, noise_px, (n_sh, 2))
        wid[rows] = pick
        descs.append(d)
        kps.append(k.astype(np.float32))
        ids.append(wid)
    return descs, kps, ids


def window_sequence(first, last, n_kp, seed, shared=0.4, window=5, sigma=6.0, noise_px=0.5, K=REF_K):
    """Images first..last-1 of an arbitrarily long capture along a straight baseline
    (the shape of an exhaustive-matching job over hundreds of views): image i
    shows pool points [i*step, i*step + n_sh), n_sh = shared * n_kp, step =
    n_sh / window, so it shares (window - d) / window of them with image i +- d
    and nothing beyond +- window.  Every pool point sits in front of the camera
    that sees it first; camera i is displaced by 12 i along x.  Each image
    depends only on (seed, i), so ranks can generate just the ones they own.
    Returns (descs, kps)."""
    n_sh = int(n_kp * shared)
    step = max(1, n_sh // window)
    Kinv = np.linalg.inv(K)
    base_x = 12.0
    descs, kps = [], []
    for i in range(first, last):
        idx = i * step + np.arange(n_sh)
        # pool point idx: descriptor, pixel in the camera that sees it first, depth -- all from its own seed block
        blk = idx // step                                  # camera that sees the point first
        d = sift_like(n_kp, seed + 100 + i)
        k_all = np.random.default_rng(seed + 7 * i + 3).uniform(0, 1, (n_kp, 2)) * np.array([IMG_W, IMG_H])
        bd = np.empty((n_sh, 128), np.float32)
        X = np.empty((n_sh, 3))
        for b in np.unique(blk):
            sel = blk == b
            r = np.random.default_rng(seed * 1000003 + int(b))
            pool_d = sift_like(step, seed * 31 + int(b) + 5)
            pool_px = np.stack([r.uniform(200, IMG_W, step), r.uniform(0, IMG_H, step), np.ones(step)], 1)
            pool_X = (pool_px @ Kinv.T) * r.uniform(425.0, 935.0, step)[:, None]
            pool_X[:, 0] += base_x * b                     # world frame: camera b sits at x = 12 b
            off = idx[sel] - b * step
            bd[sel] = pool_d[off]
            X[sel] = pool_X[off]
        rng = np.random.default_rng(seed + 13 * i + 1)
        Xc = X - np.array([base_x * i, 0.0, 0.0])
        p = Xc @ K.T
        p = p[:, :2] / p[:, 2:]
        rows = rng.permutation(n_kp)[:n_sh]
        d[rows] = np.clip(bd + np.rint(rng.normal(0.0, sigma, (n_sh, 128))), 0, 255).astype(np.float32)
        k_all[rows] = p + rng.normal(0.0, noise_px, (n_sh, 2))
        descs.append(d)
        kps.append(k_all.astype(np.float32))
    return descs, kps
