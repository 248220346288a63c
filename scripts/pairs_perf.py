"""Times match_pairs on a batch of synthetic pairs for a chosen model / mode.
usage: pairs_perf.py [model E|F] [mode adaptive|fixed] [pairs] [n_desc] [hyps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sfm_amd import matcher, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_pairs

def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "E"
    mode = sys.argv[2] if len(sys.argv) > 2 else "adaptive"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
    nh = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
    torch.cuda.init()
    descs, kps = make_pairs(B, n, 1000)
    d_desc = torch.from_numpy(np.stack(descs)).cuda()
    d_kp = torch.from_numpy(np.stack(kps)).cuda()
    blk = matcher.image_packed_bytes(n)
    d_packed = torch.empty((2 * B, blk), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = matcher.Context(0)
    matcher.prepare_many(ctx, [d_desc[i].data_ptr() for i in range(2 * B)], [n] * (2 * B), 512, matcher.DESC_F32,
                         [d_kp[i].data_ptr() for i in range(2 * B)], [d_packed[i].data_ptr() for i in range(2 * B)])
    matcher.check_inputs(ctx)
    images = [matcher.WrappedImage(ctx, d_packed[i].data_ptr(), n) for i in range(2 * B)]
    pairs = np.array([[2 * p, 2 * p + 1] for p in range(B)], np.int32)
    kw = dict(ratio=0.7, model=matcher.MODEL_ESSENTIAL if model == "E" else matcher.MODEL_FUNDAMENTAL,
              mode=matcher.RANSAC_ADAPTIVE if mode == "adaptive" else matcher.RANSAC_FIXED, prob=0.99,
              threshold=1.0 if model == "E" else 3.0, max_iters=1000, n_hyp=nh)
    K = synth.REF_K if model == "E" else None
    for _ in range(2):
        ng, ni = matcher.match_pairs(ctx, images, pairs, K, **kw)
    ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ng, ni = matcher.match_pairs(ctx, images, pairs, K, **kw)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%s %s: %d pairs of %d: %.3f ms/batch  %.0f pairs/s  good %.0f inliers %.1f" %
          (model, mode, B, n, dt * 1e3, B / dt, ng.mean(), ni.mean()))
    for name in ("knn2_mfma", "knn_finalize", "knn_redo", "ratio_compact", "ransac_prep", "ransac_table",
                 "ransac_solve", "ransac_score", "ransac_select", "ransac_mask", "inlier_compact"):
        ms, cnt = ctx.profile_query(name)
        if cnt:
            print("   %-14s %8.1f us x %d per batch" % (name, ms / cnt * 1e3, cnt // reps))
main()
