import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from sfm_amd import matcher
from bench import make_pairs
B, n = 64, 8192
torch.cuda.init()
ctx = matcher.Context(0)
for wf in (0.07, 0.25, 0.12):
    descs, kps = make_pairs(B, n, 1000 + int(wf * 100), wf)
    d_desc = torch.from_numpy(np.stack(descs)).cuda(); d_kp = torch.from_numpy(np.stack(kps)).cuda()
    d_packed = torch.empty((2 * B, matcher.image_packed_bytes(n)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    matcher.prepare_many(ctx, [d_desc[i].data_ptr() for i in range(2 * B)], [n] * (2 * B), 512, matcher.DESC_F32,
                         [d_kp[i].data_ptr() for i in range(2 * B)], [d_packed[i].data_ptr() for i in range(2 * B)])
    matcher.check_inputs(ctx)
    images = [matcher.WrappedImage(ctx, d_packed[i].data_ptr(), n) for i in range(2 * B)]
    cnts = []
    for p0 in range(0, 8):
        pairs = np.array([[2 * ((p0 + p) % B), 2 * ((p0 + p) % B) + 1] for p in range(B)], np.int32)
        matcher.match_pairs(ctx, images, pairs, None, ratio=0.7, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=64)
        c = ctypes.c_int32(0)
        ctx.check(ctx._L.sfm_gpu_debug_redo_count(ctx.handle, ctypes.byref(c)))
        cnts.append(c.value)
    print("wrong_frac", wf, "redo queries of the first job over 8 batches:", cnts)
