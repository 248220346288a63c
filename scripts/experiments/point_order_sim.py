"""How many inlier tests does bounded scoring need, by the order the points are visited in?  (CPU, oracle arithmetic.)

A model is dropped, exactly, once its misses exceed n - B (B = best complete count known).  Natural order against
"points the best-known model misses first".  point_order_sim.py [wrong_frac] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from oracle import pyoracle as po
from sfm_amd import synth

wf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.07
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_desc, n_hyp, thr = 8192, 2048, 3.0
Q, T, planted = synth.planted_pair(n_desc, n_desc, seed + 1, seed + 2, seed + 3)
kq, kt = synth.keypoints_for_pair(n_desc, n_desc, planted, seed + 4, wrong_frac=wf)
m, c = po.knn_match(Q, T, 2, nthreads=8, u8path=True)
keep = po.ratio_test(m, c, 0.7)
good = m[keep, 0]
p1 = kq[good["queryIdx"]].astype(np.float32)
p2 = kt[good["trainIdx"]].astype(np.float32)
n = len(p1)
tab = po.sample_table_fixed(n, 7, n_hyp, p1, p2)
models = []
for row in tab:
    for F in po.seven_point(p1[row], p2[row]):
        models.append(F)
M = len(models)
inl = np.zeros((M, n), bool)
t2 = np.float32(thr * thr)
for i, F in enumerate(models):
    inl[i] = po.symepi_errors(F, p1, p2) <= t2
counts = inl.sum(1)
best = counts.max()
print("n=%d models=%d best=%d (%.1f%% misses)  median count %d" % (n, M, best, 100 * (n - best) / n, np.median(counts)))

OPEN = 256


def cost(order, B):
    """tests per model when points are visited in `order` and a model stops at its (n - B + 1)-th miss"""
    miss = ~inl[:, order]
    cm = np.cumsum(miss, axis=1)
    allowed = n - B
    dead_at = np.argmax(cm > allowed, axis=1)  # index of the miss that kills it; 0 if never
    never = cm[:, -1] <= allowed
    t = np.where(never, n, dead_at + 1)
    return t


# what the pipeline knows after the opening: counts over the first 256 points; it then completes the top 2
open_counts = inl[:, :OPEN].sum(1)
top = np.argsort(-open_counts, kind="stable")[:2]
B0 = counts[top].max()
lead = top[np.argmax(counts[top])]
nat = np.arange(n)
t_nat = cost(nat, B0)
rest = np.arange(OPEN, n)
miss_first = np.concatenate([np.arange(OPEN), rest[~inl[lead, rest]], rest[inl[lead, rest]]])
t_new = cost(miss_first, B0)
t_ideal = cost(np.concatenate([np.arange(OPEN), rest[~inl[np.argmax(counts), rest]], rest[inl[np.argmax(counts), rest]]]), best)
print("bound after the opening + 2 completed models: B0 = %d" % B0)
for name, t in (("natural order", t_nat), ("best-known model's misses first", t_new), ("ideal (final best known)", t_ideal)):
    tt = np.maximum(t, OPEN)
    print("  %-34s mean points per model %7.1f  (%.1f%% of n)   tests %.1f M" % (name, tt.mean(), 100 * tt.mean() / n, tt.sum() / 1e6))
# histogram of where models die
for name, t in (("natural", t_nat), ("misses first", t_new)):
    print("  %s: died by 512: %.0f%%  by 1024: %.0f%%  by 2048: %.0f%%  survive: %.0f%%" % (
        name, 100 * (t <= 512).mean(), 100 * (t <= 1024).mean(), 100 * (t <= 2048).mean(), 100 * (t >= n).mean()))
