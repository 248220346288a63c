#!/usr/bin/env python3
"""Runs the reference's match() loop (main.cpp:159-268: ring of consecutive
pairs, kNN k=2, ratio 0.7, essential-matrix RANSAC 0.99 / 1.0 px) over a
directory of descriptor blobs (sfm_amd/blob.py), e.g. SIFT features of the
reference's data/ sequence extracted elsewhere; then the three-view joins of
main.cpp:442-456.  Prints the per-pair counts the reference logs (info.txt).

    python scripts/match_blobs.py DIR [--all-pairs] [--fundamental]
    python scripts/match_blobs.py --selftest     # writes synthetic blobs first
"""
import argparse
import glob
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def selftest_dir(n_img=6, n=4096):
    from sfm_amd import blob, synth
    d = tempfile.mkdtemp(prefix="sfm_blobs_")
    descs, kps, _ = synth.image_sequence(n_img, n, 77)
    for i in range(n_img):
        # every other image stored as CV_8U: both ingest paths in one run
        blob.write_blob(os.path.join(d, "img_%03d.sfmd" % i), descs[i].astype(np.uint8) if i % 2 else descs[i],
                        kps[i], synth.REF_K)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?")
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--all-pairs", action="store_true")
    ap.add_argument("--fundamental", action="store_true")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    from sfm_amd import allpairs, blob, matcher, synth
    d = selftest_dir() if args.selftest else args.dir
    files = sorted(glob.glob(os.path.join(d, "*.sfmd")))
    if len(files) < 2:
        raise SystemExit("need at least two blobs in %s" % d)
    ctx = matcher.Context(0)
    blobs = [blob.read_blob(f) for f in files]
    t0 = time.perf_counter()
    images = []
    for b in blobs:
        if b["keypoints"] is None:
            raise SystemExit("blobs need keypoints for RANSAC")
        images.append(matcher.Image(ctx, np.ascontiguousarray(b["descriptors"]), np.ascontiguousarray(b["keypoints"])))
    V = len(images)
    pairs = np.array(allpairs.pair_list(V, "all" if args.all_pairs else "ring"), np.int32)
    K = blobs[0]["K"] if blobs[0]["K"] is not None else synth.REF_K
    if args.fundamental:
        n_good, n_inl, sel = matcher.match_pairs(ctx, images, pairs, None, model=matcher.MODEL_FUNDAMENTAL,
                                                 prob=0.99, threshold=3.0, want_matches=True)
        poses = None
    else:
        # with the pose of every pair from its own essential matrix (main.cpp:400-420 for each pair)
        poses = matcher.match_pairs_pose(ctx, images, pairs, K, prob=0.99, threshold=1.0, want_matches=True)
        n_good, n_inl, sel = poses["n_good"], poses["n_inliers"], poses["inlier_matches"]
    t1 = time.perf_counter()
    for (i, j), g, k in zip(pairs.tolist(), n_good, n_inl):
        print("pair %3d %3d: %6d x %6d descriptors, good matches %6d, inliers %6d" %
              (i, j, images[i].n, images[j].n, g, k))
    print("%d images, %d pairs: upload + match + RANSAC %.1f ms" % (V, len(pairs), (t1 - t0) * 1e3))
    if poses is not None:
        for p, (i, j) in enumerate(pairs.tolist()):
            if poses["n_pose_good"][p] >= 0:
                print("pair %3d %3d: %d of %d inliers in front of both cameras, t = %s" %
                      (i, j, poses["n_pose_good"][p], n_inl[p], np.round(poses["t"][p], 3).tolist()))
    if not args.all_pairs:
        # the ring's lists are consecutive pairs (i, i+1), (i+1, i+2): all joins in one call
        t2 = time.perf_counter()
        tracks = matcher.track_join_sequence(ctx, sel)
        t3 = time.perf_counter()
        for i, tr in enumerate(tracks):
            if pairs[i][1] == pairs[i + 1][0]:
                print("images %d-%d-%d: %d three-view tracks" % (pairs[i][0], pairs[i][1], pairs[i + 1][1], len(tr)))
        print("%d joins in one call: %.2f ms" % (len(tracks), (t3 - t2) * 1e3))
    for im in images:
        im.close()
    ctx.close()


if __name__ == "__main__":
    main()
