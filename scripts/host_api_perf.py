"""Times the drop-in calls with HOST buffers (upload, conversion, compute,
download all inside): sfm_gpu_knn_match, sfm_gpu_ransac and sfm_gpu_recover_pose
as the reference would call them (main.cpp:184, :224, :420)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_amd import matcher, synth

def main():
    ctx = matcher.Context(0)
    bf = matcher.BFMatcher(ctx)
    for n in (8192, 20000, 32768):
        Q, T, planted = synth.planted_pair(n, n, 11, 12, 13)
        kq, kt = synth.keypoints_for_pair(n, n, planted, 14, wrong_frac=0.07)
        for _ in range(2):
            m, c = bf.knnMatch(Q, T, 2)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            m, c = bf.knnMatch(Q, T, 2)
        t_knn = (time.perf_counter() - t0) / reps
        keep = matcher.ratio_test(m, c, 0.7)
        good = m[keep, 0]
        p1 = kq[good["queryIdx"]].astype(np.float64)
        p2 = kt[good["trainIdx"]].astype(np.float64)
        for _ in range(2):
            E, mask, r = matcher.findEssentialMat(ctx, p1, p2, synth.REF_K, prob=0.99, threshold=1.0)
        t0 = time.perf_counter()
        for _ in range(reps):
            E, mask, r = matcher.findEssentialMat(ctx, p1, p2, synth.REF_K, prob=0.99, threshold=1.0)
        t_rs = (time.perf_counter() - t0) / reps
        for _ in range(2):
            matcher.recoverPose(ctx, E, p1, p2, synth.REF_K)
        t0 = time.perf_counter()
        for _ in range(reps * 4):
            ng, R, t, pm = matcher.recoverPose(ctx, E, p1, p2, synth.REF_K)
        t_pose = (time.perf_counter() - t0) / (reps * 4)
        ctx.profile_enable(True)
        ctx.profile_reset()
        matcher.recoverPose(ctx, E, p1, p2, synth.REF_K)
        k_ms, _ = ctx.profile_query("pose_vote")
        ctx.profile_enable(False)
        print("n=%d: recoverPose(%d pts) %.3f ms host call, vote kernel %.1f us, %d in front" %
              (n, len(p1), t_pose * 1e3, k_ms * 1e3, ng))
        # three-view join (main.cpp:442-456) on lists of the size this pair produced
        rng = np.random.default_rng(n)
        m3 = int(mask.sum())
        ab = np.zeros(m3, matcher.DMATCH_DTYPE); bc = np.zeros(m3, matcher.DMATCH_DTYPE)
        ab["queryIdx"] = np.sort(rng.choice(n, m3, replace=False)); ab["trainIdx"] = rng.choice(n, m3, replace=False)
        bc["queryIdx"] = np.sort(rng.choice(n, m3, replace=False)); bc["trainIdx"] = rng.choice(n, m3, replace=False)
        for _ in range(2):
            tr = matcher.track_join(ctx, ab, bc)
        t0 = time.perf_counter()
        for _ in range(reps * 4):
            tr = matcher.track_join(ctx, ab, bc)
        t_join = (time.perf_counter() - t0) / (reps * 4)
        print("n=%d: track_join(%d x %d matches) %.3f ms host call, %d tracks" % (n, m3, m3, t_join * 1e3, len(tr)))
        mb = 2 * n * 128 * 4 / 1e6
        print("n=%d: knnMatch(host f32, %.0f MB up) %.2f ms; findEssentialMat(%d pts, adaptive) %.2f ms, "
              "%d iterations, %d inliers" % (n, mb, t_knn * 1e3, len(p1), t_rs * 1e3, r["iters"], r["n_inliers"]))
main()
