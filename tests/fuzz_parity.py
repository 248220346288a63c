"""Randomised parity sweep: the GPU library against the CPU oracle on inputs
the fixed test cases do not cover (odd sizes, degenerate geometry, quantised
pixels, repeated rows, extreme parameters).  Runs for a bounded time, prints a
progress line per batch of trials, stores the inputs of every mismatch under
gpurun_out/ and exits non-zero if there was one.

    python tests/fuzz_parity.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import pyoracle as oracle          # noqa: E402  (the checker)
from sfm_amd import matcher, synth             # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out")


def _rot(rng, max_deg):
    axis = rng.normal(size=3)
    return synth._rot(axis, np.deg2rad(rng.uniform(0.0, max_deg)))


def scene(rng, n, kind):
    """Two views of one of several scene kinds.  Pixels in float32 precision."""
    K = synth.REF_K
    Kinv = np.linalg.inv(K)
    frac = rng.choice([1.0, 0.95, 0.8, 0.5, 0.25, 0.08])
    n_in = max(int(round(n * frac)), min(n, 5))
    px = np.stack([rng.uniform(0, synth.IMG_W, n_in), rng.uniform(0, synth.IMG_H, n_in), np.ones(n_in)], 1)
    depth = rng.uniform(400.0, 950.0, n_in)
    R = _rot(rng, 8.0)
    t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * rng.choice([0.05, 5.0, 25.0, 120.0])
    if kind == "planar":
        depth = 600.0 + 0.1 * (px[:, 0] - 900.0)           # one slanted plane
    elif kind == "rotation":
        t = np.zeros(3)                                      # no parallax at all
    elif kind == "forward":
        t = np.array([0.0, 0.0, 60.0])
        R = _rot(rng, 0.3)
    elif kind == "line":
        px[:, 1] = 0.4 * px[:, 0] + 100.0                    # every point on one image line
    X = (px @ Kinv.T) * np.asarray(depth).reshape(-1, 1)
    X2 = X @ R.T + t
    p2 = X2 @ K.T
    p2 = p2[:, :2] / p2[:, 2:]
    noise = rng.choice([0.0, 0.3, 1.0])
    p1 = px[:, :2] + rng.normal(0, noise, (n_in, 2))
    p2 = p2 + rng.normal(0, noise, (n_in, 2))
    n_out = n - n_in
    o = rng.uniform(0, 1, (n_out, 4)) * [synth.IMG_W, synth.IMG_H, synth.IMG_W, synth.IMG_H]
    pts1 = np.concatenate([p1, o[:, :2]], 0)
    pts2 = np.concatenate([p2, o[:, 2:]], 0)
    perm = rng.permutation(n)
    pts1, pts2 = pts1[perm], pts2[perm]
    if kind == "integer":
        pts1, pts2 = np.rint(pts1), np.rint(pts2)           # many equal coordinates
    if kind == "repeats" and n > 8:
        src = rng.integers(0, n, n // 2)
        dst = rng.integers(0, n, n // 2)
        pts1[dst], pts2[dst] = pts1[src], pts2[src]
    return pts1.astype(np.float32).astype(np.float64), pts2.astype(np.float32).astype(np.float64)


KINDS = ["general", "general", "planar", "rotation", "forward", "line", "integer", "repeats"]


def save_failure(tag, idx, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "fuzz_fail_%s_%d.npz" % (tag, idx))
    np.savez_compressed(path, **arrays)
    return path


def close(a, b, tol=1e-6):
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)


def trial_ransac(ctx, rng, idx, stats):
    model = int(rng.integers(0, 2))
    m = 5 if model == 0 else 7
    n = int(rng.choice([m, m + 1, 9, 33, 64, 65, 200, 511, 513, 1000, 2500, 6000]))
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    p1, p2 = scene(rng, n, kind)
    K = synth.REF_K if model == 0 else None
    thr = float(rng.choice([0.1, 1.0, 3.0, 10.0]))
    prob = float(rng.choice([0.5, 0.99, 0.999, 0.999999]))
    fixed = bool(rng.integers(0, 2))
    if fixed:
        nh = int(rng.choice([1, 2, 31, 64, 100, 257, 600]))
        o = oracle.ransac(p1, p2, K, model=model, prob=prob, threshold=thr, fixed_count=nh)
        r = matcher._ransac(ctx, p1, p2, K, model, prob, thr, 1000, matcher.RANSAC_FIXED, None, nh, None)
        what = "fixed nh=%d" % nh
    else:
        mi = int(rng.choice([1, 10, 100, 1000, 2000]))
        if model == 1 and 7 < n < 15:
            # cv::findFundamentalMat runs LMedS here: the library must refuse, not run RANSAC
            try:
                matcher._ransac(ctx, p1, p2, K, model, prob, thr, mi, matcher.RANSAC_ADAPTIVE, None, 0, None)
            except matcher.SfmGpuError as e:
                stats["refused"] = stats.get("refused", 0) + 1
                return 0 if e.code == -5 else 1
            print("MISMATCH ransac #%d: fundamental adaptive with n=%d was not refused" % (idx, n), flush=True)
            return 1
        o = oracle.ransac(p1, p2, K, model=model, prob=prob, threshold=thr, max_iters=mi)
        r = matcher._ransac(ctx, p1, p2, K, model, prob, thr, mi, matcher.RANSAC_ADAPTIVE, None, 0, None)
        what = "adaptive max_iters=%d" % mi
    bad = []
    if r["ok"] != o["ok"]:
        bad.append("ok %d vs %d" % (r["ok"], o["ok"]))
    elif o["ok"]:
        if r["n_inliers"] != o["n_inliers"] or not (r["mask"] == o["mask"]).all():
            bad.append("mask/inliers %d vs %d" % (r["n_inliers"], o["n_inliers"]))
        if not fixed and r["iters"] != o["iters"]:
            bad.append("iters %d vs %d" % (r["iters"], o["iters"]))
        if not close(r["model"], o["model"]):
            bad.append("model off by %.3g" % np.abs(r["model"] - o["model"]).max())
        elif not (r["model"] == o["model"]).all():
            stats["model_last_bits"] += 1
    stats["ransac"] += 1
    stats["ransac_ok"] += int(o["ok"])
    if bad:
        path = save_failure("ransac", idx, p1=p1, p2=p2, model=model, thr=thr, prob=prob)
        print("MISMATCH ransac #%d model=%d n=%d %s %s thr=%g prob=%g: %s -> %s"
              % (idx, model, n, kind, what, thr, prob, "; ".join(bad), path), flush=True)
        return 1
    # pose from the essential matrix just found, restricted to its inliers
    if model == 0 and o["ok"]:
        dist = float(rng.choice([50.0, 5.0, 1e6]))
        use_mask = o["mask"] if rng.integers(0, 2) else None
        po = oracle.recover_pose(o["model"], p1, p2, synth.REF_K, dist, use_mask)
        try:
            good, R, t, pm = matcher.recoverPose(ctx, o["model"], p1, p2, synth.REF_K, dist, use_mask)
        except Exception as e:                                # the library reports failure as an error
            good, R, t, pm = -1, None, None, None
            if po["n_good"] >= 0:
                print("MISMATCH pose #%d: library raised %r, oracle n_good=%d" % (idx, e, po["n_good"]), flush=True)
                save_failure("pose", idx, E=o["model"], p1=p1, p2=p2)
                return 1
        stats["pose"] += 1
        if good >= 0:
            pbad = []
            if good != po["n_good"] or not (pm == po["mask"]).all():
                pbad.append("n_good %d vs %d" % (good, po["n_good"]))
            if not close(R, po["R"]) or not close(t.reshape(3), po["t"]):
                pbad.append("R/t differ")
            if pbad:
                path = save_failure("pose", idx, E=o["model"], p1=p1, p2=p2, dist=dist,
                                    mask=np.zeros(0) if use_mask is None else use_mask)
                print("MISMATCH pose #%d n=%d %s: %s -> %s" % (idx, n, kind, "; ".join(pbad), path), flush=True)
                return 1
    return 0


def descriptors(rng, n, kind):
    if kind == "sift":
        return synth.sift_like(n, int(rng.integers(1, 1 << 30)))
    if kind == "uniform":
        return rng.integers(0, 256, (n, 128)).astype(np.float32)
    if kind == "coarse":
        return (rng.integers(0, 3, (n, 128)) * 60).astype(np.float32)      # many equal distances
    if kind == "sparse":
        d = np.zeros((n, 128), np.float32)
        d[np.arange(n), rng.integers(0, 128, n)] = rng.integers(0, 256, n)
        return d
    return np.full((n, 128), float(rng.integers(0, 256)), np.float32)      # all rows equal


def trial_knn(ctx, bf, rng, idx, stats):
    nq = int(rng.choice([1, 2, 63, 64, 65, 500, 1023, 2049, 4000]))
    nt = int(rng.choice([1, 2, 3, 127, 128, 129, 1000, 4097, 9000]))
    kind = ["sift", "uniform", "coarse", "sparse", "constant"][int(rng.integers(0, 5))]
    Q = descriptors(rng, nq, kind)
    T = descriptors(rng, nt, kind)
    if nq > 4 and nt > 4 and rng.integers(0, 2):
        rows = rng.integers(0, nt, min(nq, nt) // 2)
        T[rows] = Q[rng.integers(0, nq, len(rows))]                      # exact copies, some repeated
    k = int(rng.choice([1, 2, 2, 2]))
    as_u8 = bool(rng.integers(0, 2))
    if as_u8:
        gm, gc = bf.knnMatch(Q.astype(np.uint8), T.astype(np.uint8), k)
    else:
        gm, gc = bf.knnMatch(Q, T, k)
    om, oc = oracle.knn_match(Q, T, k, nthreads=8)
    stats["knn"] += 1
    ok = (gc == oc).all()
    if ok:
        for j in range(k):
            live = oc > j
            ok = ok and (gm[live, j]["trainIdx"] == om[live, j]["trainIdx"]).all() \
                and (gm[live, j]["queryIdx"] == om[live, j]["queryIdx"]).all() \
                and (gm[live, j]["distance"].view(np.uint32) == om[live, j]["distance"].view(np.uint32)).all()
    if not ok:
        path = save_failure("knn", idx, Q=Q, T=T, k=k)
        print("MISMATCH knn #%d nq=%d nt=%d k=%d %s u8=%d -> %s" % (idx, nq, nt, k, kind, as_u8, path), flush=True)
        return 1
    return 0


def trial_pairs(ctx, rng, idx, stats):
    """The whole pipeline on a few small ragged images."""
    n_img = int(rng.integers(2, 5))
    sizes = [int(rng.choice([8, 40, 300, 777, 1500, 2600])) for _ in range(n_img)]
    pool = synth.sift_like(3000, int(rng.integers(1, 1 << 30)))
    descs, kps = [], []
    for n in sizes:
        d = pool[rng.permutation(3000)[:n]]
        d = np.clip(d + np.rint(rng.normal(0, rng.choice([0.0, 3.0, 12.0]), d.shape)), 0, 255).astype(np.float32)
        descs.append(d)
        kps.append(np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32))
    pairs = np.array([[i, j] for i in range(n_img) for j in range(n_img) if i != j], np.int32)
    pairs = pairs[rng.permutation(len(pairs))[:int(rng.integers(1, len(pairs) + 1))]]
    ratio = float(rng.choice([0.6, 0.7, 0.8, 0.95]))
    model = int(rng.integers(0, 2))
    nh = int(rng.choice([16, 64, 200]))
    thr = 3.0 if model else 1.0
    K = None if model else synth.REF_K
    fixed = bool(rng.integers(0, 2))
    as_u8 = bool(rng.integers(0, 2))
    imgs = [matcher.Image(ctx, d.astype(np.uint8) if as_u8 else d, k) for d, k in zip(descs, kps)]
    kw = dict(ratio=ratio, model=model, mode=matcher.RANSAC_FIXED if fixed else matcher.RANSAC_ADAPTIVE,
              prob=0.99, threshold=thr, n_hyp=nh if fixed else 0, max_iters=300)
    try:
        ng, ni, sel = matcher.match_pairs(ctx, imgs, pairs, K, want_matches=True, **kw)
        if rng.integers(0, 2):
            # the same batch through the enqueue / collect entry points, twice in flight
            # (half of the time with each pair's records left on the device and copied to pinned host memory)
            rec = bool(rng.integers(0, 2))
            cap = max(sizes)
            if rec:
                import torch
                from sfm_amd import _lib
                dev = [torch.full((len(pairs), cap, 4), -7, dtype=torch.int32, device="cuda") for _ in range(2)]
                host = [torch.full((len(pairs), cap, 4), -9, dtype=torch.int32).pin_memory() for _ in range(2)]
                torch.cuda.synchronize()
                bs = [matcher.PairBatch(ctx, imgs, pairs, K, records_dev=d.data_ptr(), match_capacity=cap,
                                        records_host=h.data_ptr(), **kw) for d, h in zip(dev, host)]
            else:
                bs = [matcher.PairBatch(ctx, imgs, pairs, K, **kw)] * 2
            bs[0].begin()
            bs[1].begin()
            for t in range(2):
                g2, i2 = bs[t].end()
                if not ((g2 == ng).all() and (i2 == ni).all()):
                    print("MISMATCH pairs #%d: begin/end counts differ from the blocking call" % idx, flush=True)
                    return 1
                if rec:
                    ctx.synchronize()
                    h = host[t].numpy().view(_lib.DMATCH_DTYPE).reshape(len(pairs), cap)
                    d = dev[t].cpu().numpy().view(_lib.DMATCH_DTYPE).reshape(len(pairs), cap)
                    for p in range(len(pairs)):
                        if not (h[p, :ni[p]].tobytes() == sel[p].tobytes() and d[p, :ni[p]].tobytes() == sel[p].tobytes()):
                            print("MISMATCH pairs #%d: begin/end records differ from the blocking call's" % idx, flush=True)
                            return 1
    finally:
        for im in imgs:
            im.close()
    stats["pairs"] += len(pairs)
    for p, (i, j) in enumerate(pairs):
        m, c = oracle.knn_match(descs[i], descs[j], 2, nthreads=8)
        good = m[oracle.ratio_test(m, c, ratio), 0]
        p1 = kps[i][good["queryIdx"]].astype(np.float64)
        p2 = kps[j][good["trainIdx"]].astype(np.float64)
        kw = dict(fixed_count=nh) if fixed else dict(max_iters=300)
        r = oracle.ransac(p1, p2, K, model=model, prob=0.99, threshold=thr, **kw)
        osel = good[r["mask"].astype(bool)] if r["ok"] else good[:0]
        same = ng[p] == len(good) and ni[p] == len(osel) and (sel[p] == osel).all()
        if not same:
            path = save_failure("pairs", idx, dq=descs[i], dt=descs[j], kq=kps[i], kt=kps[j], ratio=ratio,
                                model=model)
            print("MISMATCH pairs #%d pair (%d,%d) sizes %s ratio=%g model=%d fixed=%d: good %d/%d inl %d/%d -> %s"
                  % (idx, i, j, sizes, ratio, model, fixed, ng[p], len(good), ni[p], len(osel), path), flush=True)
            return 1
    return 0


def trial_tracks(ctx, rng, idx, stats):
    na = int(rng.choice([0, 1, 50, 3000]))
    nb = int(rng.choice([0, 1, 70, 4000]))
    kp = int(rng.choice([5, 200, 5000]))
    ab = np.zeros(na, matcher.DMATCH_DTYPE)
    ab["queryIdx"] = np.sort(rng.integers(0, kp, na))
    ab["trainIdx"] = rng.integers(0, kp, na)
    bc = np.zeros(nb, matcher.DMATCH_DTYPE)
    bc["queryIdx"] = np.sort(rng.integers(0, kp, nb))     # ascending, with repeats
    bc["trainIdx"] = rng.integers(0, kp, nb)
    got = matcher.track_join(ctx, ab, bc)
    want = oracle.track_join(ab, bc)
    stats["tracks"] += 1
    if got.shape != want.shape or not (got == want).all():
        path = save_failure("tracks", idx, ab=ab, bc=bc)
        print("MISMATCH tracks #%d na=%d nb=%d -> %s" % (idx, na, nb, path), flush=True)
        return 1
    return 0


def trial_sequence(ctx, rng, idx, stats):
    """Joins of a whole sequence in one call against the loop per image."""
    n_lists = int(rng.choice([0, 1, 2, 3, 7]))
    kp = int(rng.choice([5, 300, 6000]))
    ls = []
    for l in range(n_lists):
        n = int(rng.choice([0, 1, 40, 1024, 2500]))
        m = np.zeros(n, matcher.DMATCH_DTYPE)
        m["queryIdx"] = np.sort(rng.integers(0, kp, n))
        m["trainIdx"] = rng.integers(0, kp, n)
        ls.append(m)
    got = matcher.track_join_sequence(ctx, ls)
    stats["sequence"] += 1
    bad = len(got) != max(n_lists - 1, 0)
    for l in range(n_lists - 1):
        want = oracle.track_join(ls[l], ls[l + 1])
        bad = bad or got[l].shape != want.shape or not (got[l] == want).all()
    if bad:
        path = save_failure("sequence", idx, **{"l%d" % l: m for l, m in enumerate(ls)})
        print("MISMATCH sequence #%d lists %s -> %s" % (idx, [len(m) for m in ls], path), flush=True)
        return 1
    return 0


def trial_pnp(ctx, rng, idx, stats):
    """solvePnPRansac on planted poses: general, planar and repeated object
    points, 0-90 % wrong correspondences, odd parameters."""
    n = int(rng.choice([5, 6, 7, 12, 60, 400, 3000]))
    kind = str(rng.choice(["general", "planar", "repeated"]))
    X = np.stack([rng.uniform(-300, 300, n), rng.uniform(-200, 200, n), rng.uniform(500, 1100, n)], 1)
    if kind == "planar":
        X[:, 2] = 800.0 + 0.05 * X[:, 0]
    if kind == "repeated" and n > 6:
        X[n // 2:] = X[:n - n // 2]
    R = _rot(rng, 25.0)
    t = np.array([rng.normal(0, 30), rng.normal(0, 30), rng.normal(0, 60)])
    Xc = X @ R.T + t
    u = (Xc @ synth.REF_K.T)
    u = u[:, :2] / u[:, 2:]
    u += rng.normal(0, float(rng.choice([0.0, 0.3, 2.0])), u.shape)
    wrong = rng.uniform(size=n) < float(rng.choice([0.0, 0.1, 0.4, 0.9]))
    u[wrong] = np.stack([rng.uniform(0, synth.IMG_W, wrong.sum()), rng.uniform(0, synth.IMG_H, wrong.sum())], 1)
    iters = int(rng.choice([1, 10, 100, 300]))
    err = float(rng.choice([0.5, 2.0, 8.0, 30.0]))
    conf = float(rng.choice([0.5, 0.99, 0.999999]))
    ok, rvec, tvec, _, info = matcher.solvePnPRansac(ctx, X, u, synth.REF_K, iterationsCount=iters,
                                                     reprojectionError=err, confidence=conf)
    o = oracle.pnp_ransac(X, u, synth.REF_K, iterations_count=iters, reprojection_error=err, confidence=conf)
    stats["pnp"] += 1
    same = bool(ok) == bool(o["ok"])
    if same and o["ok"]:
        same = (info["iters"] == o["iters"] and info["n_inliers"] == o["n_inliers"] and
                (info["mask"] == o["mask"]).all() and info["R"].tobytes() == o["R"].tobytes() and
                tvec.tobytes() == o["tvec"].tobytes() and rvec.tobytes() == o["rvec"].tobytes())
    if not same:
        path = save_failure("pnp", idx, X=X, u=u, iters=iters, err=err, conf=conf)
        print("MISMATCH pnp #%d n=%d %s iters=%d err=%g conf=%g: ok %d/%d -> %s"
              % (idx, n, kind, iters, err, conf, ok, o["ok"], path), flush=True)
        return 1
    return 0


def run(seconds, seed, only="", ctx=None, quiet=False, score_prune=None):
    """Draws trials for `seconds`; returns (fails, trials, stats).
    score_prune: scoring schedule of the fixed-count mode to force on the context
    (sfm_gpu_debug_set_score_prune; 4 = the ordered stages for any batch size)."""
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        import torch
        torch.cuda.init()   # before the first context (the pipeline trials keep record buffers in torch tensors)
        ctx = matcher.Context(0)
    if score_prune is not None:
        ctx.check(ctx._L.sfm_gpu_debug_set_score_prune(ctx.handle, int(score_prune)))
    bf = matcher.BFMatcher(ctx)
    stats = dict(ransac=0, ransac_ok=0, model_last_bits=0, pose=0, knn=0, pairs=0, tracks=0, sequence=0, pnp=0)
    only = [s for s in only.split(",") if s]
    menu = [("ransac", 6), ("knn", 2), ("pairs", 1), ("tracks", 1), ("sequence", 0.5), ("pnp", 1)]
    menu = [(k, w) for k, w in menu if not only or k in only]
    names = [k for k, _ in menu]
    weights = np.array([w for _, w in menu], float)
    weights /= weights.sum()
    fails = 0
    t0 = time.time()
    last = t0
    idx = 0
    while time.time() - t0 < seconds:
        what = names[int(rng.choice(len(names), p=weights))]
        if what == "ransac":
            fails += trial_ransac(ctx, rng, idx, stats)
        elif what == "knn":
            fails += trial_knn(ctx, bf, rng, idx, stats)
        elif what == "pairs":
            fails += trial_pairs(ctx, rng, idx, stats)
        elif what == "sequence":
            fails += trial_sequence(ctx, rng, idx, stats)
        elif what == "pnp":
            fails += trial_pnp(ctx, rng, idx, stats)
        else:
            fails += trial_tracks(ctx, rng, idx, stats)
        idx += 1
        if not quiet and time.time() - last > 20:
            last = time.time()
            print("[%4.0fs] trials=%d fails=%d %s" % (last - t0, idx, fails, stats), flush=True)
    if score_prune is not None:
        ctx.check(ctx._L.sfm_gpu_debug_set_score_prune(ctx.handle, 1))
    if own:
        ctx.close()
    return fails, idx, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="", help="comma list of ransac,knn,pairs,tracks,sequence,pnp")
    ap.add_argument("--score-prune", type=int, default=None,
                    help="force a scoring schedule of the fixed-count mode (4: ordered stages for any batch)")
    args = ap.parse_args()
    fails, idx, stats = run(args.seconds, args.seed, args.only, score_prune=args.score_prune)
    print("done: trials=%d fails=%d %s" % (idx, fails, stats), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
