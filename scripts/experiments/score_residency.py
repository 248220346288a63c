"""Where and when the wavefronts of the scoring launches run (timing build: per-wavefront
stamps): per launch tag the number of wavefronts, their lifetimes, how many share a SIMD at a
time and how evenly the SIMDs are loaded.
usage: score_residency.py [wrong_frac]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from sfm_amd import _lib
_lib.LIB_PATH = _lib.TIMING_LIB_PATH   # before the first call: this process runs on the timing build
from sfm_amd import matcher
from bench import make_pairs

wf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.07
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
N = 1 << 18
buf = np.zeros((N, 4), np.uint64)
cnt = ctypes.c_int32(0)
for _ in range(3):
    matcher.match_pairs(ctx, images, pairs, None, **kw)
ctx.check(ctx._L.sfm_gpu_debug_score_stamps(ctx.handle, buf.ctypes.data, N, ctypes.byref(cnt)))   # clears
matcher.match_pairs(ctx, images, pairs, None, **kw)
ctx.check(ctx._L.sfm_gpu_debug_score_stamps(ctx.handle, buf.ctypes.data, N, ctypes.byref(cnt)))
st = buf[:cnt.value]
st = st[st[:, 1] != 0]   # slots are interleaved over 64 counters: unused ones are zero
print("stamps:", len(st))
tags = (st[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.uint32).astype(np.int32).astype(np.int64)
dA = ((st[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0   # entry -> models known to be live
dB = ((st[:, 2] >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0   # entry -> model and filter set up
dC = ((st[:, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0   # entry -> float32 loop done
cyc = ((st[:, 3] >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.float64) * 16.0    # shader cycles of the wavefront's life
for tag in sorted(set(tags.tolist())):
    s = st[tags == tag]
    t0, t1 = s[:, 0].astype(np.int64), s[:, 1].astype(np.int64)
    base = t0.min()
    hw = s[:, 2]
    # gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13; XCC id in the upper word
    simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
    xcc = ((hw >> np.uint64(16)) & np.uint64(15)).astype(np.int64)
    unit = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    life = (t1 - t0) / 100.0
    span = (t1.max() - base) / 100.0
    per = np.bincount(np.unique(unit, return_inverse=True)[1])
    # concurrency on a SIMD: time-weighted mean of resident wavefronts while it has any
    conc = []
    for u in np.unique(unit)[:200]:
        m = unit == u
        ev = sorted([(a, 1) for a in t0[m]] + [(b, -1) for b in t1[m]])
        cur, last, acc, busy = 0, ev[0][0], 0.0, 0.0
        for t, d in ev:
            if cur > 0:
                acc += cur * (t - last)
                busy += t - last
            cur += d
            last = t
        conc.append(acc / max(busy, 1))
    print("tag %3d: %6d wavefronts, launch span %.1f us, starts %.1f..%.1f us (median %.1f), lifetime median %.1f us "
          "(p10 %.1f, p90 %.1f), SIMDs used %d, wavefronts per SIMD min/median/max %d/%d/%d, resident while busy %.1f"
          % (tag, len(s), span, 0.0, (t0.max() - base) / 100.0, np.median(t0 - base) / 100.0, np.median(life),
             np.percentile(life, 10), np.percentile(life, 90), len(per), per.min(), np.median(per), per.max(),
             float(np.mean(conc))))
    sel = tags == tag
    print("         phases (median us from entry): live check %.1f, model set up %.1f, float32 loop done %.1f, end %.1f"
          % (np.median(dA[sel]), np.median(dB[sel]), np.median(dC[sel]), np.median(life)))
    print("         shader clock over the wavefronts' lives: median %.0f MHz" % np.median(cyc[sel] / life))
