"""Ablation ladder of the distance sweep (knn2_mfma16), timing build only.

Times the same launch with parts of the kernel switched off (wrong results),
variants interleaved in rounds inside one process so that clock and box effects
hit all of them alike:

    full | no digest | no DMA | no barrier | no DMA, no barrier | MFMA + LDS reads only

for (a) the bench batch, 64 pairs of 8192 x 8192, and (b) one 32768 x 32768 pair.
Prints one JSON document; usage: sweep_ladder.py [rounds] [out.json]
"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sfm_amd import matcher, synth
from bench import make_pairs

MASKS = [(0, "full"), (1, "no digest"), (2, "no DMA of further stages"), (4, "no stage barrier"),
         (6, "no DMA, no barrier"), (3, "no digest, no DMA"), (7, "MFMAs + LDS fragment reads only")]
PEAK = 5.0e15
SHAPE = int(os.environ.get('SWEEP_SHAPE', '4'))


def ladder(run, ctx, ops, rounds, reps):
    per = {m: [] for m, _ in MASKS}
    for m, _ in MASKS:          # warm every variant's code
        ctx._L.sfm_gpu_debug_set_knn_variant(ctx.handle, SHAPE + 16 * m)
        run()
    ctx.synchronize()
    for _ in range(rounds):
        for m, _ in MASKS:
            ctx.check(ctx._L.sfm_gpu_debug_set_knn_variant(ctx.handle, SHAPE + 16 * m))
            ctx.profile_reset()
            for _ in range(reps):
                run()
            ms, cnt = ctx.profile_query("knn2_mfma")
            per[m].append(ms / cnt * 1e3)
    ctx._L.sfm_gpu_debug_set_knn_variant(ctx.handle, -1)
    out = []
    for m, name in MASKS:
        med = statistics.median(per[m])
        out.append({"mask": m, "what": name, "median_us": round(med, 2), "min_us": round(min(per[m]), 2),
                    "frac_of_nominal_peak": round(ops / (med * 1e-6) / PEAK, 4)})
    return out


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    torch.cuda.init()
    ctx = matcher.Context(0, timing_build=True)
    ctx.profile_select("knn2_mfma")
    ctx.profile_enable(True)
    res = {"shape": SHAPE, "library": ctx._L.sfm_gpu_version().decode(), "rounds": rounds,
           "method": "HIP events around each launch, variants interleaved per round, median over rounds"}

    # (a) the bench batch
    B, n = 64, 8192
    descs, kps = make_pairs(B, n, 1000, 0.07)
    d_desc = torch.from_numpy(np.stack(descs)).cuda()
    d_kp = torch.from_numpy(np.stack(kps)).cuda()
    blk = matcher.image_packed_bytes(n)
    d_packed = torch.empty((2 * B, blk), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    matcher.prepare_many(ctx, [d_desc[i].data_ptr() for i in range(2 * B)], [n] * (2 * B), 512, matcher.DESC_F32,
                         [d_kp[i].data_ptr() for i in range(2 * B)], [d_packed[i].data_ptr() for i in range(2 * B)])
    matcher.check_inputs(ctx)
    images = [matcher.WrappedImage(ctx, d_packed[i].data_ptr(), n) for i in range(2 * B)]
    pairs = np.array([[2 * p, 2 * p + 1] for p in range(B)], np.int32)

    def run_batch():
        matcher.match_pairs(ctx, images, pairs, None, ratio=0.7, model=matcher.MODEL_FUNDAMENTAL,
                            mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=2048)

    res["batch_64x8192"] = ladder(run_batch, ctx, 2.0 * n * n * 128 * B, rounds, 3)
    for im in images:
        im.close()
    del d_desc, d_kp, d_packed

    # (b) one 32768 x 32768 pair
    n = 32768
    Q, T, _ = synth.planted_pair(n, n, 1101, 1102, 1103)
    iq, it = matcher.Image(ctx, Q), matcher.Image(ctx, T)
    out = torch.empty(n * 2 * 16, dtype=torch.uint8, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")

    def run_one():
        matcher.knn_match_images(ctx, iq, it, 2, (out.data_ptr(), cnt.data_ptr()))

    res["lone_32768"] = ladder(run_one, ctx, 2.0 * n * n * 128, rounds, 10)
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(res, f, indent=1)
    ctx.close()


main()
