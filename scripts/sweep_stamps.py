"""Timeline of one sweep launch from in-kernel stamps (timing build): how long
the workgroups spend before, inside and after their stage loop, and how their
start and end times spread.  usage: sweep_stamps.py [n (default 32768)] [shape]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sfm_amd import matcher, synth

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    shape = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    torch.cuda.init()
    ctx = matcher.Context(0, timing_build=True)
    Q, T, _ = synth.planted_pair(n, n, 1101, 1102, 1103)
    iq, it = matcher.Image(ctx, Q), matcher.Image(ctx, T)
    out = torch.empty(n * 2 * 16, dtype=torch.uint8, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.check(ctx._L.sfm_gpu_debug_set_knn_variant(ctx.handle, shape))
    for _ in range(20):
        matcher.knn_match_images(ctx, iq, it, 2, (out.data_ptr(), cnt.data_ptr()))
    ctx.synchronize()
    nwg = 512 if shape == 4 else 256
    st = np.zeros((nwg, 8), np.uint64)
    ctx.check(ctx._L.sfm_gpu_debug_sweep_stamps(ctx.handle, matcher.ptr(st), nwg))
    st = st.astype(np.int64)
    r0 = st[:, 0].min()
    start_us = (st[:, 0] - r0) / 100.0
    end_us = (st[:, 5] - r0) / 100.0
    pro, loop, epi = st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4] - st[:, 3]
    clk = (st[:, 4] - st[:, 1]) / ((st[:, 5] - st[:, 0]) / 100.0) / 1e3
    q = lambda a: "min %.1f med %.1f max %.1f" % (a.min(), np.median(a), a.max())
    print("workgroups", nwg, "clock GHz", q(clk))
    print("start (us after first)   ", q(start_us))
    print("end   (us after first)   ", q(end_us))
    cm = np.median(clk) * 1e3
    print("prologue us  ", q(pro / cm), " cycles med %d" % np.median(pro))
    print("stage loop us", q(loop / cm), " cycles med %d" % np.median(loop))
    print("epilogue us  ", q(epi / cm), " cycles med %d" % np.median(epi))
    # does the position in the grid tell which workgroups run ahead?
    lu = loop / cm
    ids = np.arange(nwg)
    print("loop us by grid half:   first %.1f  second %.1f" % (lu[ids < nwg // 2].mean(), lu[ids >= nwg // 2].mean()))
    print("loop us by id parity:   even %.1f  odd %.1f" % (lu[ids % 2 == 0].mean(), lu[ids % 2 == 1].mean()))
    print("loop us by (id/8) parity:  even %.1f  odd %.1f" % (lu[(ids // 8) % 2 == 0].mean(), lu[(ids // 8) % 2 == 1].mean()))
    print("loop us by eighth of the grid:", " ".join("%.1f" % lu[(ids * 8) // nwg == e].mean() for e in range(8)))
    print("fast (< median) workgroups per eighth:", " ".join("%d" % (lu[(ids * 8) // nwg == e] < np.median(lu)).sum() for e in range(8)))
    for x in range(8):
        sel = st[:, 6] == x
        if sel.any():
            print("  xcc %d: wgs %3d  end med %.1f max %.1f  prologue med %.1f us loop med %.1f us" %
                  (x, sel.sum(), np.median(end_us[sel]), end_us[sel].max(), np.median(pro[sel]) / cm, np.median(loop[sel]) / cm))
main()
