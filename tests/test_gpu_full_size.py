"""Full-size and randomised GPU parity: the benchmark's own configuration
against the CPU oracle, a bounded slice of the randomised sweep, and the
exhaustive matcher at 20 000 descriptors per view."""
import os
import sys

import numpy as np
import pytest

from sfm_amd import allpairs, matcher, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _oracle_pair(oracle, dq, dt, kq, kt, K, ratio, model, **kw):
    m, c = oracle.knn_match(dq, dt, 2, nthreads=16)
    keep = oracle.ratio_test(m, c, ratio)
    good = m[keep, 0]
    p1 = kq[good["queryIdx"]].astype(np.float64)
    p2 = kt[good["trainIdx"]].astype(np.float64)
    r = oracle.ransac(p1, p2, K, model=model, **kw)
    sel = good[r["mask"].astype(bool)] if r["ok"] else good[:0]
    return len(good), sel


def test_bench_configuration_full_size_against_oracle(gpu_ctx, oracle):
    """The headline benchmark configuration as bench.py runs it: pairs of 8192 x 8192 float32
    descriptors, ratio 0.7, fundamental RANSAC with 2048 hypotheses (fixed count),
    two pairs of the bench batch through sfm_gpu_match_pairs.  Survivor counts,
    inlier counts and every selected cv::DMatch record (distance bits included)
    must equal the oracle pipeline's."""
    sys.path.insert(0, ROOT)
    from bench import make_pairs
    descs, kps = make_pairs(2, 8192, 1000, 0.07)
    imgs = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    pairs = np.array([[0, 1], [2, 3]], np.int32)
    ng, ni, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, None, ratio=0.7, model=matcher.MODEL_FUNDAMENTAL,
                                      mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=2048,
                                      want_matches=True)
    for p in range(2):
        og, osel = _oracle_pair(oracle, descs[2 * p], descs[2 * p + 1], kps[2 * p], kps[2 * p + 1], None, 0.7,
                                oracle.MODEL_FUNDAMENTAL, prob=0.99, threshold=3.0, fixed_count=2048)
        assert ng[p] == og and ni[p] == len(osel)
        assert sel[p].tobytes() == osel.astype(matcher.DMATCH_DTYPE).tobytes()
        assert 3000 < og < 3600 and len(osel) > 0.85 * og      # the planted structure is there
    for im in imgs:
        im.close()


def test_randomised_parity_slice(gpu_ctx):
    """25 seconds of tests/fuzz_parity.py (RANSAC in both models and modes on
    degenerate scenes, recoverPose, ragged kNN, the pair pipeline, the track
    join) with a fixed seed; no trial may differ from the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fuzz_parity
    fails, trials, stats = fuzz_parity.run(25.0, 20261111, ctx=gpu_ctx, quiet=True)
    assert fails == 0, stats
    assert trials > 200 and stats["ransac"] > 50 and stats["knn"] > 10 and stats["pairs"] > 5, stats


@pytest.mark.parametrize("own_stream", [True, False])
def test_exhaustive_matcher_20000_descriptors_cuda_inputs(gpu_ctx, oracle, own_stream):
    """sfm_amd.allpairs with the GPU backend on 6 views x 20 000 descriptors (the
    per-view size of the exhaustive benchmark configuration) whose descriptors are CUDA tensors
    produced by asynchronous torch work right before the call, once with the
    library on a stream of its own and once on torch's stream.  Three of the 15
    pairs are checked against the oracle pipeline."""
    import torch
    V, n = 6, 20000
    descs, kps, _ = synth.image_sequence(V, n, 4242, shared=0.4)
    st = torch.cuda.Stream()
    ctx = matcher.Context(0) if own_stream else matcher.Context(0, stream=st.cuda_stream)
    be = allpairs.GpuBackend(ctx, K=synth.REF_K, model=matcher.MODEL_ESSENTIAL, mode=matcher.RANSAC_ADAPTIVE,
                             prob=0.99, threshold=1.0, max_iters=1000)
    with torch.cuda.stream(st):
        junk = torch.randn(4096, 4096, device="cuda")
        for _ in range(8):
            junk = junk @ junk * 1e-3          # keeps torch's stream busy ahead of the uploads' consumers
        local = []
        for d, k in zip(descs, kps):
            # values arrive through arithmetic on the stream, not through a synchronous copy
            dd = torch.from_numpy(d).cuda(non_blocking=True)
            local.append(((dd * 2.0 - dd).contiguous(), (torch.from_numpy(k).cuda(non_blocking=True) + 0.0)))
        pairs, good, inl = allpairs.match_all_pairs(local, V, be, mode="all", device=torch.device("cuda", 0))
    assert len(pairs) == 15
    for (i, j) in ((0, 1), (2, 5), (3, 4)):
        og, osel = _oracle_pair(oracle, descs[i], descs[j], kps[i], kps[j], synth.REF_K, 0.7,
                                oracle.MODEL_ESSENTIAL, prob=0.99, threshold=1.0, max_iters=1000)
        assert good[i, j] == og and inl[i, j] == len(osel), (i, j, good[i, j], og, inl[i, j], len(osel))
    assert inl[0, 1] > 1000 and good[j, i] == 0
    ctx.close()


def test_exhaustive_matcher_many_small_views(gpu_ctx, oracle):
    """The shape of the largest exhaustive configuration scaled down in rows, not
    in structure: 150 views, all 11 175 pairs, many launch groups of 256 pairs.
    The count matrices equal pair-by-pair library calls on a sample and the
    oracle pipeline on a few pairs; neighbouring views share the most inliers."""
    V, n = 150, 320
    descs, kps, _ = synth.image_sequence(V, n, 515, shared=0.6)
    be = allpairs.GpuBackend(gpu_ctx, K=synth.REF_K, model=matcher.MODEL_ESSENTIAL, mode=matcher.RANSAC_ADAPTIVE,
                             prob=0.99, threshold=1.0, max_iters=300)
    local = [(d, k) for d, k in zip(descs, kps)]
    pairs, good, inl = allpairs.match_all_pairs(local, V, be, mode="all")
    assert len(pairs) == V * (V - 1) // 2 and good.shape == (V, V)
    assert (np.tril(good) == 0).all() and (np.tril(inl) == 0).all()
    images = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    rng = np.random.default_rng(3)
    sample = [(0, 1), (0, V - 1), (V - 2, V - 1)] + [tuple(sorted(rng.choice(V, 2, replace=False))) for _ in range(40)]
    for (i, j) in sample:
        g1, i1 = matcher.match_pairs(gpu_ctx, images, np.array([[i, j]], np.int32), synth.REF_K, prob=0.99,
                                     threshold=1.0, max_iters=300)
        assert good[i, j] == g1[0] and inl[i, j] == i1[0], (i, j)
    for (i, j) in sample[:4]:
        og, osel = _oracle_pair(oracle, descs[i], descs[j], kps[i], kps[j], synth.REF_K, 0.7,
                                oracle.MODEL_ESSENTIAL, prob=0.99, threshold=1.0, max_iters=300)
        assert good[i, j] == og and inl[i, j] == len(osel), (i, j)
    # the first views look at the scene from nearby places: every neighbouring pair finds a geometry
    adj = np.array([inl[i, i + 1] for i in range(12)])
    assert adj.min() > 20
    order, parent = allpairs.spanning_order(inl)
    assert sorted(order.tolist()) == list(range(V))
    for im in images:
        im.close()


def test_long_capture_of_8192_descriptor_views_sampled_against_oracle(gpu_ctx, oracle):
    """The shape of the 1000-image exhaustive job at a size a test can afford:
    200 views x 8192 descriptors of a long capture (each view overlaps its four
    neighbours on either side), all 19 900 pairs through the exhaustive matcher
    with the reference's own RANSAC call (essential, adaptive, 0.99, 1 px); 20
    pairs -- 12 overlapping ones and 8 that share nothing -- are redone by the
    oracle pipeline, survivor and inlier counts must agree; so must the band
    structure of the whole count matrix."""
    V, n = 200, 8192
    descs, kps = synth.window_sequence(0, V, n, 4242, shared=0.4, window=5)
    images = list(zip(descs, kps))
    be = allpairs.GpuBackend(gpu_ctx, K=synth.REF_K, model=matcher.MODEL_ESSENTIAL, mode=matcher.RANSAC_ADAPTIVE,
                             prob=0.99, threshold=1.0, max_iters=1000)
    pairs, good, inl = allpairs.match_all_pairs(images, V, be, mode="all")
    assert len(pairs) == V * (V - 1) // 2
    i, j = pairs[:, 0], pairs[:, 1]
    near = (j - i) <= 4
    assert (inl[i[near], j[near]] > 300).all()         # every overlapping pair found its geometry
    assert (inl[i[~near & (j - i > 5)], j[~near & (j - i > 5)]] < 30).all()
    rng = np.random.default_rng(5)
    sample = list(rng.permutation(np.flatnonzero(near))[:12]) + list(rng.permutation(np.flatnonzero(j - i > 5))[:8])
    for p in sample:
        a, b = int(i[p]), int(j[p])
        m, c = oracle.knn_match(descs[a], descs[b], 2, nthreads=16, u8path=True)
        g = m[oracle.ratio_test(m, c, 0.7), 0]
        assert good[a, b] == len(g), (a, b)
        p1 = kps[a][g["queryIdx"]].astype(np.float64)
        p2 = kps[b][g["trainIdx"]].astype(np.float64)
        r = oracle.ransac(p1, p2, synth.REF_K, model=0, prob=0.99, threshold=1.0, max_iters=1000)
        assert inl[a, b] == (int(r["n_inliers"]) if r["ok"] else 0), (a, b)
