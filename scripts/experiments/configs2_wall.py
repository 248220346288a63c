"""Host-clock time of the one-pair call (32768 x 32768 descriptors, five-point RANSAC with 4096
hypotheses), with and without the per-launch HIP events, call by call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from sfm_amd import matcher, synth

n2, nh2 = 32768, 4096
torch.cuda.init()
Q2, T2, planted2 = synth.planted_pair(n2, n2, 1101, 1102, 1103)
kq2, kt2 = synth.keypoints_for_pair(n2, n2, planted2, 2101, wrong_frac=0.07)
d2 = torch.from_numpy(np.stack([Q2, T2])).cuda()
k2 = torch.from_numpy(np.stack([kq2, kt2])).cuda()
pk2 = torch.empty((2, matcher.image_packed_bytes(n2)), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx = matcher.Context(0)
matcher.prepare_many(ctx, [d2[i].data_ptr() for i in range(2)], [n2, n2], 512, matcher.DESC_F32,
                     [k2[i].data_ptr() for i in range(2)], [pk2[i].data_ptr() for i in range(2)])
matcher.check_inputs(ctx)
im2 = [matcher.WrappedImage(ctx, pk2[i].data_ptr(), n2) for i in range(2)]
one = np.array([[0, 1]], np.int32)
kw2 = dict(ratio=0.7, model=matcher.MODEL_ESSENTIAL, mode=matcher.RANSAC_FIXED, prob=0.99, threshold=1.0, n_hyp=nh2)
for prof in (False, True, False):
    ctx.profile_enable(prof)
    ctx.profile_reset()
    ts = []
    for _ in range(8):
        t = time.perf_counter()
        matcher.match_pairs(ctx, im2, one, synth.REF_K, **kw2)
        ctx.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    print("events", prof, " ms per call:", " ".join("%.3f" % v for v in ts))
