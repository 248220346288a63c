"""Experiment: does the sweep's speed depend on the operand bit patterns?"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sfm_amd import matcher, synth

def main():
    torch.cuda.init()
    ctx = matcher.Context(0)
    ctx.profile_select("knn2_mfma"); ctx.profile_enable(True)
    n = 32768
    rng = np.random.default_rng(1)
    sift_q, sift_t, _ = synth.planted_pair(n, n, 1101, 1102, 1103)
    cases = {
        "sift-like": (sift_q, sift_t),
        "uniform random": (rng.integers(0, 256, (n, 128)).astype(np.float32), rng.integers(0, 256, (n, 128)).astype(np.float32)),
        # the same rows moved by a constant (clipped): stored bytes are value - 128, so +128 makes the stored
        # operands small numbers around zero instead of numbers around -128
        "sift+128 both": (np.minimum(sift_q + 128, 255), np.minimum(sift_t + 128, 255)),
        "sift q, sift+128 t": (sift_q, np.minimum(sift_t + 128, 255)),
        "sift+128 q, sift t": (np.minimum(sift_q + 128, 255), sift_t),
        "sift+64 both": (np.minimum(sift_q + 64, 255), np.minimum(sift_t + 64, 255)),
        "q=0 t=0": (np.zeros((n, 128), np.float32), np.zeros((n, 128), np.float32)),
        "q=255 t=0": (np.full((n, 128), 255, np.float32), np.zeros((n, 128), np.float32)),
        "q=0 t=255": (np.zeros((n, 128), np.float32), np.full((n, 128), 255, np.float32)),
        "q=255 t=255": (np.full((n, 128), 255, np.float32), np.full((n, 128), 255, np.float32)),
        "q=128 t=128": (np.full((n, 128), 128, np.float32), np.full((n, 128), 128, np.float32)),
        "q=127 t=128": (np.full((n, 128), 127, np.float32), np.full((n, 128), 128, np.float32)),
    }
    out = torch.empty(n * 2 * 16, dtype=torch.uint8, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    imgs = {k: (matcher.Image(ctx, q), matcher.Image(ctx, t)) for k, (q, t) in cases.items()}
    per = {k: [] for k in cases}
    for rnd in range(6):
        for k, (iq, it) in imgs.items():
            ctx.profile_reset()
            for _ in range(4):
                matcher.knn_match_images(ctx, iq, it, 2, (out.data_ptr(), cnt.data_ptr()))
            ctx.synchronize()
            ms, c = ctx.profile_query("knn2_mfma")
            if rnd: per[k].append(ms / c * 1e3)
    for k in cases:
        print("%-16s sweep %.1f us" % (k, statistics.median(per[k])), flush=True)

main()
