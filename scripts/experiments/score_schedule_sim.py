"""Counts the (model, point) tests different scoring schedules would do on one
fixed-count fundamental-matrix problem, from the oracle's models and errors:
the exhaustive loop, the per-model ideal (a model stops at the first multiple
of 128 points at which it can no longer reach the winner's count), wavefronts
of 64 models ordered by an opening stretch (a wavefront runs until its last
model stops), and the same without ordering.
    python scripts/experiments/score_schedule_sim.py [n_points] [inlier_frac] [hyps]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import pyoracle as po  # noqa: E402
from sfm_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3276
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.93
nh = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
p1, p2, _, _ = synth.two_view(n, 77, inlier_frac=frac)
f1, f2 = p1.astype(np.float32), p2.astype(np.float32)
table = po.sample_table_fixed(n, 7, nh, f1, f2)
models = []
for row in table:
    for F in po.seven_point(f1[row], f2[row]):
        models.append(F)
thr2 = np.float32(3.0 * 3.0)
inl = np.stack([po.symepi_errors(F, f1, f2) <= thr2 for F in models])  # [models, points]
nm = len(models)
full = inl.sum(1)
B = full.max()
print("models %d, points %d, winner %d (%.1f %%), allowed misses %d" % (nm, n, B, 100.0 * B / n, n - B))
cum = np.cumsum(inl, 1)
ends = np.minimum(np.arange(128, n + 128, 128), n)
# death time per model: first boundary e with cum[e-1] + n - e < B
alive_after = cum[:, ends - 1] + (n - ends)[None, :] >= B
death = np.full(nm, n)
for m in range(nm):
    dead = np.nonzero(~alive_after[m])[0]
    if len(dead):
        death[m] = ends[dead[0]]
print("exhaustive      %.2f M tests" % (nm * n / 1e6))
print("ideal per model %.2f M tests" % (death.sum() / 1e6))
for open_pts in (128, 256):
    order = np.argsort(-cum[:, open_pts - 1], kind="stable")
    for label, o in (("sorted by opening %d" % open_pts, order), ("unsorted", np.arange(nm))):
        tot = 0
        for g in range(0, nm, 64):
            tot += 64 * max(death[o[g:g + 64]].max(), open_pts)
        print("waves of 64, %-22s %.2f M tests" % (label, tot / 1e6))
hist = np.bincount(np.searchsorted(ends, death), minlength=len(ends))
print("models stopping at each boundary:", dict((int(ends[i]), int(c)) for i, c in enumerate(hist) if c))
# how good a bound do the best K models of the opening stretch give?
for open_pts in (128, 256):
    order = np.argsort(-cum[:, open_pts - 1], kind="stable")
    for K in (1, 8, 16, 64, 128):
        bK = full[order[:K]].max()
        al = cum[:, ends - 1] + (n - ends)[None, :] >= bK
        d = np.full(nm, n)
        for m in range(nm):
            dd = np.nonzero(~al[m])[0]
            if len(dd):
                d[m] = ends[dd[0]]
        tot = sum(64 * max(d[order[g:g + 64]].max(), open_pts) for g in range(0, nm, 64))
        print("opening %d, bound from best %3d: B = %d, sorted waves %.2f M tests" % (open_pts, K, bK, tot / 1e6))
# breadth-first stages over the ordered list: at each boundary a wavefront of 64 goes on if any of its models can
# still reach the bound
order = np.argsort(-cum[:, 255], kind="stable")
for bounds in ([256, 512, 1024, 2048], [256, 384, 512, 768, 1024, 1536, 2048], [256, 384, 512, 640, 896, 1280, 1792, 2432],
               [256, 512, 768, 1280, 2048], [256, 384, 640, 1024, 1664, 2432]):
    bb = [b for b in bounds if b < n] + [n]
    tot = nm * bb[0]
    detail = []
    for s, e_ in zip(bb[:-1], bb[1:]):
        alive = cum[:, s - 1] + (n - s) >= B
        waves = sum(1 for g in range(0, nm, 64) if alive[order[g:g + 64]].any())
        tot += waves * 64 * (e_ - s)
        detail.append(waves)
    print("stages at %-52s %.2f M tests, live waves per stage %s" % (bb, tot / 1e6, detail))
