#!/usr/bin/env python3
"""Writes the fixtures in this directory.

The reference repository holds no test vectors for the matching path and its
arithmetic lives in OpenCV, which is not installed here, so these fixtures are
NOT outputs of the reference: they are outputs of the CPU oracle (oracle/) on
seeded synthetic inputs, frozen so that (a) a change in the oracle shows up as
a diff against history and (b) the GPU path is checked against bytes that do
not depend on the oracle being rebuilt.  Inputs are stored too, so the files
are self-contained.

    python tests/golden/make_golden.py [knn] [essential] [fundamental] [pose] [tracks]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import pyoracle as po
from sfm_amd import synth


def knn_case():
    Q, T, planted = synth.planted_pair(300, 421, 11, 12, 13)
    # a few exact duplicates (distance ties -> ascending train index) and a far row
    T[7] = T[3]
    T[400] = T[3]
    Q[5] = T[3]
    Q[6] = 0
    T[9] = 255
    m, c = po.knn_match(Q, T, 2)
    keep = po.ratio_test(m, c, 0.7)
    np.savez_compressed(os.path.join(HERE, "knn_small.npz"), query=Q.astype(np.uint8), train=T.astype(np.uint8),
                        trainIdx=m["trainIdx"], distance=m["distance"], counts=c, keep=keep)


def ransac_case(name, model, n, seed, hyps):
    p1, p2, _, _ = synth.two_view(n, seed, 0.6, 0.4)
    K = synth.REF_K
    Kuse = K if model == po.MODEL_ESSENTIAL else None
    thr = 1.0 if model == po.MODEL_ESSENTIAL else 3.0
    fixed = po.ransac(p1, p2, Kuse, model=model, prob=0.99, threshold=thr, fixed_count=hyps, want_counts=True)
    adaptive = po.ransac(p1, p2, Kuse, model=model, prob=0.99, threshold=thr, max_iters=300)
    np.savez_compressed(
        os.path.join(HERE, name), pts1=p1, pts2=p2, K=K, threshold=thr, hyps=hyps,
        fixed_ok=fixed["ok"], fixed_mask=fixed["mask"], fixed_model=fixed["model"],
        fixed_n_inliers=fixed["n_inliers"], fixed_counts=fixed["counts"],
        adaptive_ok=adaptive["ok"], adaptive_mask=adaptive["mask"], adaptive_model=adaptive["model"],
        adaptive_n_inliers=adaptive["n_inliers"], adaptive_iters=adaptive["iters"])


def pose_case():
    """The chain of main.cpp:418-420: essential matrix by RANSAC, then the pose."""
    p1, p2, _, _ = synth.two_view(600, 23, 0.7, 0.4)
    K = synth.REF_K
    e = po.ransac(p1, p2, K, model=po.MODEL_ESSENTIAL, prob=0.999, threshold=1.0)
    assert e["ok"] == 1
    free = po.recover_pose(e["model"], p1, p2, K)
    masked = po.recover_pose(e["model"], p1, p2, K, mask=e["mask"])
    near = po.recover_pose(e["model"], p1, p2, K, distance_thresh=25.0)
    np.savez_compressed(
        os.path.join(HERE, "pose_essential.npz"), pts1=p1, pts2=p2, K=K, E=e["model"], ransac_mask=e["mask"],
        R=free["R"], t=free["t"], mask=free["mask"], votes=free["votes"], n_good=free["n_good"],
        masked_mask=masked["mask"], masked_votes=masked["votes"],
        near_mask=near["mask"], near_votes=near["votes"])


def tracks_case():
    """Inlier matches of two consecutive pairs of a three-image ring, joined as
    at main.cpp:442-456 (plus two repeated query rows in the second list)."""
    rng = np.random.default_rng(31)
    n_kp = 900
    ab = np.zeros(420, dtype=po.DMATCH_DTYPE)
    ab["queryIdx"] = np.sort(rng.choice(n_kp, 420, replace=False))
    ab["trainIdx"] = rng.choice(n_kp, 420, replace=False)
    bc = np.zeros(380, dtype=po.DMATCH_DTYPE)
    q = np.sort(rng.choice(n_kp, 380, replace=False))
    q[7] = q[6]
    q[200] = q[199]
    bc["queryIdx"] = q
    bc["trainIdx"] = rng.choice(n_kp, 380, replace=False)
    t = po.track_join(ab, bc)
    np.savez_compressed(os.path.join(HERE, "track_join.npz"), ab_query=ab["queryIdx"], ab_train=ab["trainIdx"],
                        bc_query=bc["queryIdx"], bc_train=bc["trainIdx"], tracks=t)


def pnp_case():
    """Resection (bundle_adjustment.h:139-155): 600 object points seen by a planted
    camera, 25 % of the pixels replaced by random ones, oracle_pnp_ransac defaults."""
    rng = np.random.default_rng(41)
    K = synth.REF_K
    R = synth._rot(rng.normal(size=3), np.deg2rad(17.0))
    t = np.array([12.0, -20.0, 640.0])
    X = rng.uniform(-150, 150, (600, 3))
    Xc = X @ R.T + t
    u = Xc @ K.T
    u = u[:, :2] / u[:, 2:] + rng.normal(0, 0.6, (600, 2))
    u[:150] = rng.uniform(0, 1500, (150, 2))
    perm = rng.permutation(600)
    X, u = X[perm], u[perm]
    o = po.pnp_ransac(X, u, K)
    assert o["ok"] == 1
    np.savez_compressed(os.path.join(HERE, "pnp_ransac.npz"), X=X, u=u, R=o["R"], rvec=o["rvec"], tvec=o["tvec"],
                        mask=o["mask"], iters=o["iters"], n_inliers=o["n_inliers"], R_true=R, t_true=t)


def main():
    which = sys.argv[1:] or ["knn", "essential", "fundamental", "pose", "tracks", "pnp"]
    if "pnp" in which:
        pnp_case()
    if "knn" in which:
        knn_case()
    if "essential" in which:
        ransac_case("ransac_essential.npz", po.MODEL_ESSENTIAL, 400, 21, 96)
    if "fundamental" in which:
        ransac_case("ransac_fundamental.npz", po.MODEL_FUNDAMENTAL, 500, 22, 128)
    if "pose" in which:
        pose_case()
    if "tracks" in which:
        tracks_case()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
