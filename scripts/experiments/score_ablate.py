"""Scoring time of a fixed-count batch with parts of the scoring switched off
(timing build only: modes 2 and 3 give WRONG counts).
    mode 1: the product's scoring; 2: tests the float32 filter cannot settle are
    dropped instead of redone in double precision; 3: also no point loads in the loop.
usage: score_ablate.py [wrong_frac] [prune_mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from sfm_amd import _lib
_lib.LIB_PATH = _lib.TIMING_LIB_PATH   # before the first call: this process runs on the timing build
from sfm_amd import matcher
from bench import make_pairs

wf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.07
pm = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, n, nh = 64, 8192, 2048
torch.cuda.init()
descs, kps = make_pairs(B, n, 1000, wf)
d_desc = torch.from_numpy(np.stack(descs)).cuda()
d_kp = torch.from_numpy(np.stack(kps)).cuda()
d_packed = torch.empty((2 * B, matcher.image_packed_bytes(n)), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx = matcher.Context(0)
matcher.prepare_many(ctx, [d_desc[i].data_ptr() for i in range(2 * B)], [n] * (2 * B), 512, matcher.DESC_F32,
                     [d_kp[i].data_ptr() for i in range(2 * B)], [d_packed[i].data_ptr() for i in range(2 * B)])
matcher.check_inputs(ctx)
images = [matcher.WrappedImage(ctx, d_packed[i].data_ptr(), n) for i in range(2 * B)]
pairs = np.array([[2 * p, 2 * p + 1] for p in range(B)], np.int32)
kw = dict(ratio=0.7, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=nh)
ctx.check(ctx._L.sfm_gpu_debug_set_score_prune(ctx.handle, pm))
for mode in (1, 2, 3, 1):
    ctx.check(ctx._L.sfm_gpu_debug_set_score_filter(ctx.handle, mode))
    for _ in range(2):
        ng, ni = matcher.match_pairs(ctx, images, pairs, None, **kw)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(5):
        ng, ni = matcher.match_pairs(ctx, images, pairs, None, **kw)
    ctx.synchronize()
    ms, cnt = ctx.profile_query("ransac_score")
    print("wrong %.2f prune %d filter mode %d: ransac_score %.1f us per batch, inliers %.1f" %
          (wf, pm, mode, ms / cnt * 1e3, ni.mean()))
    ctx.profile_enable(False)
