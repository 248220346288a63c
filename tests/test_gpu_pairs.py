"""End-to-end pair matching (kNN -> ratio -> RANSAC -> inlier matches) on the
GPU against the same pipeline assembled from the CPU oracle."""
import numpy as np
import pytest

from sfm_amd import matcher, synth

pytestmark = pytest.mark.gpu


def _make_images(n_img, n_desc, seed):
    descs, kps = [], []
    base = synth.sift_like(n_desc, seed)
    rng = np.random.default_rng(seed)
    for v in range(n_img):
        descs.append(base.copy())
        kps.append(None)
    # chain: image v+1 holds noisy copies of 45% of image v's rows
    out_d, out_k = [], []
    Q = synth.sift_like(n_desc, seed)
    for v in range(n_img):
        out_d.append(Q)
        Tn, Tnext, planted = synth.planted_pair(n_desc, n_desc, seed + 10 * v, seed + 10 * v + 1,
                                                seed + 10 * v + 2, frac=0.45)
        out_k.append((Tn, Tnext, planted))
    return out_d, out_k


def _oracle_pair(oracle, dq, dt, kq, kt, K, ratio, model, **kw):
    m, c = oracle.knn_match(dq, dt, 2, nthreads=8)
    keep = oracle.ratio_test(m, c, ratio)
    good = m[keep, 0]
    p1 = kq[good["queryIdx"]].astype(np.float64)
    p2 = kt[good["trainIdx"]].astype(np.float64)
    r = oracle.ransac(p1, p2, K, model=model, **kw)
    sel = good[r["mask"].astype(bool)] if r["ok"] else good[:0]
    return len(good), sel


@pytest.mark.parametrize("mode", ["adaptive", "fixed"])
def test_match_pairs_ring_like_reference(gpu_ctx, oracle, mode):
    # three images in a ring, pairs (i, (i+1) % n) like main.cpp:167-173
    n_img, n_desc = 3, 2500
    descs, kps = [], []
    for v in range(n_img):
        descs.append(synth.sift_like(n_desc, 300 + v))
    planted_all = []
    rng = np.random.default_rng(5)
    kps = [np.stack([rng.uniform(0, synth.IMG_W, n_desc), rng.uniform(0, synth.IMG_H, n_desc)],
                    1).astype(np.float32) for _ in range(n_img)]
    for v in range(n_img):
        w = (v + 1) % n_img
        rows_q = rng.permutation(n_desc)[:900]
        rows_t = rng.permutation(n_desc)[:900]
        if v == n_img - 1:
            # do not overwrite rows of image 0 that pair 0 planted from
            pass
        noise = np.rint(rng.normal(0, 6, (900, 128)))
        descs[w] = descs[w].copy()
        descs[w][rows_t] = np.clip(descs[v][rows_q] + noise, 0, 255).astype(np.float32)
        planted = np.full(n_desc, -1, np.int64)
        planted[rows_q] = rows_t
        kq, kt = synth.keypoints_for_pair(n_desc, n_desc, planted, 700 + v)
        kps[v][rows_q] = kq[rows_q]
        kps[w][rows_t] = kt[rows_t]
    imgs = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    pairs = np.array([[i, (i + 1) % n_img] for i in range(n_img)], np.int32)
    kw = dict(prob=0.99, threshold=1.0)
    if mode == "fixed":
        n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, synth.REF_K, ratio=0.7,
                                                 mode=matcher.RANSAC_FIXED, n_hyp=200,
                                                 want_matches=True, **kw)
        okw = dict(fixed_count=200, **kw)
    else:
        n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, synth.REF_K, ratio=0.7,
                                                 want_matches=True, **kw)
        okw = dict(max_iters=1000, **kw)
    for p, (i, j) in enumerate(pairs):
        ng, osel = _oracle_pair(oracle, descs[i], descs[j], kps[i], kps[j], synth.REF_K, 0.7, 0, **okw)
        assert n_good[p] == ng
        assert n_inl[p] == len(osel)
        assert (sel[p]["queryIdx"] == osel["queryIdx"]).all()
        assert (sel[p]["trainIdx"] == osel["trainIdx"]).all()
        assert (sel[p]["distance"].view(np.uint32) == osel["distance"].view(np.uint32)).all()
        assert n_inl[p] > 200
    for im in imgs:
        im.close()


def test_match_pairs_fundamental_and_ragged_sizes(gpu_ctx, oracle):
    sizes = [1800, 700, 2300]
    rng = np.random.default_rng(17)
    descs = [synth.sift_like(n, 900 + i) for i, n in enumerate(sizes)]
    kps = [np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
           for n in sizes]
    # plant image0 -> image2 and image1 -> image2 overlaps
    for (a, b, cnt, seed) in [(0, 2, 600, 1), (1, 2, 300, 2)]:
        rq = rng.permutation(sizes[a])[:cnt]
        rt = rng.permutation(sizes[b])[:cnt]
        descs[b][rt] = np.clip(descs[a][rq] + np.rint(rng.normal(0, 6, (cnt, 128))), 0, 255).astype(np.float32)
        planted = np.full(sizes[a], -1, np.int64)
        planted[rq] = rt
        kq, kt = synth.keypoints_for_pair(sizes[a], sizes[b], planted, 50 + seed)
        kps[a][rq] = kq[rq]
        kps[b][rt] = kt[rt]
    imgs = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    pairs = np.array([[0, 2], [1, 2], [2, 1]], np.int32)
    n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, None, ratio=0.7,
                                             model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED,
                                             n_hyp=256, prob=0.99, threshold=3.0, want_matches=True)
    for p, (i, j) in enumerate(pairs):
        ng, osel = _oracle_pair(oracle, descs[i], descs[j], kps[i], kps[j], None, 0.7, 1,
                                prob=0.99, threshold=3.0, fixed_count=256)
        assert n_good[p] == ng and n_inl[p] == len(osel)
        assert (sel[p]["trainIdx"] == osel["trainIdx"]).all()
    for im in imgs:
        im.close()


def test_allpairs_driver_single_process_equals_match_pairs(gpu_ctx):
    """sfm_amd.allpairs with the GPU backend (world size 1: same code path as a
    rank of a multi-GPU job minus the two collectives) against direct
    match_pairs calls on owned images."""
    from sfm_amd import allpairs
    sizes = [1500, 2048, 1100, 1900]
    rng = np.random.default_rng(9)
    pool = synth.sift_like(2048, 41)
    images = []
    for n in sizes:
        d = pool[rng.permutation(2048)[:n]].copy()
        d = np.clip(d + rng.integers(-4, 5, d.shape), 0, 255).astype(np.float32)
        kp = np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
        images.append((d, kp))
    kw = dict(K=None, ratio=0.7, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99,
              threshold=3.0, n_hyp=64)
    pairs, good, inl = allpairs.match_all_pairs(images, len(images), allpairs.GpuBackend(gpu_ctx, **kw),
                                                mode="all")
    assert pairs.tolist() == allpairs.pair_list(4, "all").tolist()
    owned = [matcher.Image(gpu_ctx, d, k) for d, k in images]
    ng, ni = matcher.match_pairs(gpu_ctx, owned, pairs, None, **{k: v for k, v in kw.items() if k != "K"})
    for (i, j), g, n in zip(pairs, ng, ni):
        assert good[i, j] == g and inl[i, j] == n
    assert good.sum() == ng.sum() and inl.sum() == ni.sum() and ng.sum() > 0
    # the capture-order step that consumes these matrices: a spanning tree over all four images
    order, parent = allpairs.spanning_order(inl)
    assert sorted(order.tolist()) == [0, 1, 2, 3] and (parent >= 0).sum() == 3
    for im in owned:
        im.close()


def test_match_pairs_many_small_pairs_in_two_groups_and_bad_index(gpu_ctx):
    """More pairs than one launch group holds (256): the last, smaller group is
    planned differently and must give the same answers as pair-by-pair calls."""
    from sfm_amd import _lib
    rng = np.random.default_rng(31)
    pool = synth.sift_like(600, 51)
    images = []
    for v in range(12):
        n = 300 + 17 * v
        d = np.clip(pool[rng.permutation(600)[:n]] + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.float32)
        kp = np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
        images.append(matcher.Image(gpu_ctx, d, kp))
    pairs = np.array([[i, j] for i in range(12) for j in range(12) if i != j] * 2, np.int32)[:261]
    kw = dict(ratio=0.8, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99,
              threshold=3.0, n_hyp=32)
    ng, ni, sel = matcher.match_pairs(gpu_ctx, images, pairs, None, want_matches=True, **kw)
    for p in (0, 100, 255, 256, 260):
        g1, i1, s1 = matcher.match_pairs(gpu_ctx, images, pairs[p:p + 1], None, want_matches=True, **kw)
        assert g1[0] == ng[p] and i1[0] == ni[p]
        assert (s1[0] == sel[p]).all()
    assert ng.sum() > 0
    with pytest.raises(_lib.SfmGpuError) as e:
        matcher.match_pairs(gpu_ctx, images, np.array([[0, 12]], np.int32), None, **kw)
    assert e.value.code == _lib.ERR_ARG
    for im in images:
        im.close()


@pytest.mark.parametrize("ratio", [0.3, 0.6, 0.8, 0.95, 0.999, 1.0, 1.5])
def test_ratio_screening_never_changes_the_survivors(gpu_ctx, oracle, ratio):
    """The pair pipeline gives no exact neighbours to queries that certainly fail
    the ratio test; whatever the ratio, the survivors (and what RANSAC makes of
    them) are those of the oracle's knnMatch + ratio test on every query.  The
    descriptor noise is spread so that d1/d2 covers the whole range around each
    tested ratio."""
    n = 3000
    rng = np.random.default_rng(int(ratio * 1000))
    Q = synth.sift_like(n, 311)
    T = synth.sift_like(n, 312)
    rows_q = rng.permutation(n)[:1500]
    rows_t = rng.permutation(n)[:1500]
    sigma = rng.uniform(2.0, 40.0, (1500, 1))          # from sure matches to indistinguishable ones
    T[rows_t] = np.clip(Q[rows_q] + np.rint(rng.normal(0, 1, (1500, 128)) * sigma), 0, 255).astype(np.float32)
    T[5] = T[6]                                          # exact duplicates: d1 == d2 for their match
    planted = np.full(n, -1, np.int64)
    planted[rows_q] = rows_t
    kq, kt = synth.keypoints_for_pair(n, n, planted, 313)
    imgs = [matcher.Image(gpu_ctx, Q, kq), matcher.Image(gpu_ctx, T, kt)]
    pairs = np.array([[0, 1], [1, 0]], np.int32)
    n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, None, ratio=ratio, model=matcher.MODEL_FUNDAMENTAL,
                                             mode=matcher.RANSAC_FIXED, n_hyp=64, prob=0.99, threshold=3.0,
                                             want_matches=True)
    for p, (dq, dt, k1, k2) in enumerate([(Q, T, kq, kt), (T, Q, kt, kq)]):
        ng, osel = _oracle_pair(oracle, dq, dt, k1, k2, None, ratio, 1, prob=0.99, threshold=3.0, fixed_count=64)
        assert n_good[p] == ng and n_inl[p] == len(osel)
        assert (sel[p]["queryIdx"] == osel["queryIdx"]).all() and (sel[p]["trainIdx"] == osel["trainIdx"]).all()
        assert (sel[p]["distance"].view(np.uint32) == osel["distance"].view(np.uint32)).all()
    if ratio <= 0.95:
        assert 0 < n_good[0] < n
    for im in imgs:
        im.close()


@pytest.mark.parametrize("ratio", [0.5, 0.8, 1.0])
def test_pipeline_on_tie_heavy_descriptors(gpu_ctx, oracle, ratio):
    """Low-entropy descriptors: many equal distances, many group-minimum ties.
    The pipeline resolves a tie without the exhaustive pass only when neither
    the best match nor the ratio verdict can depend on it; the survivors are
    still exactly the oracle's."""
    n = 1500
    rng = np.random.default_rng(7 + int(ratio * 10))
    Q = (rng.integers(0, 3, (n, 128)) * 40).astype(np.float32)
    T = (rng.integers(0, 3, (n, 128)) * 40).astype(np.float32)
    rows = rng.permutation(n)[:700]
    T[rows] = Q[rng.permutation(n)[:700]]            # exact copies, some of them repeated
    T[rows[:50]] = T[rows[50:100]]
    flip = rng.integers(0, 128, 700)
    T[rows, flip] = 80.0 - T[rows, flip]              # one coordinate off for most copies
    kq = np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
    kt = np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
    imgs = [matcher.Image(gpu_ctx, Q, kq), matcher.Image(gpu_ctx, T, kt)]
    pairs = np.array([[0, 1], [1, 0]], np.int32)
    n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, None, ratio=ratio, model=matcher.MODEL_FUNDAMENTAL,
                                             mode=matcher.RANSAC_FIXED, n_hyp=32, prob=0.99, threshold=1e6,
                                             want_matches=True)
    for p, (dq, dt) in enumerate([(Q, T), (T, Q)]):
        m, c = oracle.knn_match(dq, dt, 2, nthreads=4)
        good = m[oracle.ratio_test(m, c, ratio), 0]
        assert n_good[p] == len(good)
        if n_inl[p] == len(good):                      # threshold 1e6: every survivor is an inlier
            assert (sel[p]["queryIdx"] == good["queryIdx"]).all() and (sel[p]["trainIdx"] == good["trainIdx"]).all()
            assert (sel[p]["distance"].view(np.uint32) == good["distance"].view(np.uint32)).all()
    assert n_good[0] > 50
    for im in imgs:
        im.close()


def test_verdicts_that_hang_on_one_unit_of_squared_distance(gpu_ctx, oracle):
    """The pipeline's neighbour extraction computes exact distances for the best
    group of 16 train rows only and knows the second distance up to one unit
    (d2 = L or L + 1) unless it looks closer.  Here every query's two nearest rows
    sit in different groups at d1^2 = 100 and d2^2 = 400 or 401, and the ratio 0.5
    passes exactly the latter: 10 < 0.5 * sqrt(401) but not 10 < 0.5 * 20."""
    n_q, n_t = 700, 2400
    rng = np.random.default_rng(77)
    Q = synth.sift_like(n_q, 501)
    T = synth.sift_like(n_t, 502)
    want_pass = rng.integers(0, 2, n_q).astype(bool)
    for i in range(n_q):
        small = np.argsort(Q[i])[:3]                   # three coordinates with room to grow
        near, far = Q[i].copy(), Q[i].copy()
        near[small[0]] += 10.0                          # d^2 = 100
        far[small[1]] += 20.0                           # d^2 = 400
        if want_pass[i]:
            far[small[2]] += 1.0                        # d^2 = 401
        T[i] = near
        T[n_q + 16 + ((i * 37) % (n_t - n_q - 16))] = far   # another group of 16 rows, scattered
    # (a later query's far row may overwrite an earlier one's: the oracle sees the same images)
    kq = np.stack([rng.uniform(0, synth.IMG_W, n_q), rng.uniform(0, synth.IMG_H, n_q)], 1).astype(np.float32)
    kt = np.stack([rng.uniform(0, synth.IMG_W, n_t), rng.uniform(0, synth.IMG_H, n_t)], 1).astype(np.float32)
    imgs = [matcher.Image(gpu_ctx, Q, kq), matcher.Image(gpu_ctx, T, kt)]
    pairs = np.array([[0, 1]] * 8, np.int32)
    n_good, n_inl, sel = matcher.match_pairs(gpu_ctx, imgs, pairs, None, ratio=0.5, model=matcher.MODEL_FUNDAMENTAL,
                                             mode=matcher.RANSAC_FIXED, n_hyp=32, prob=0.99, threshold=1e6,
                                             want_matches=True)
    m, c = oracle.knn_match(Q, T, 2, nthreads=4)
    good = m[oracle.ratio_test(m, c, 0.5), 0]
    d2 = np.rint(m["distance"].astype(np.float64) ** 2)
    assert ((d2[:, 0] == 100) & ((d2[:, 1] == 400) | (d2[:, 1] == 401))).sum() > 500   # the construction held
    assert 100 < len(good) < n_q - 100
    for p in range(len(pairs)):
        assert n_good[p] == len(good)
        if n_inl[p] == len(good):
            assert (sel[p]["queryIdx"] == good["queryIdx"]).all() and (sel[p]["trainIdx"] == good["trainIdx"]).all()
            assert (sel[p]["distance"].view(np.uint32) == good["distance"].view(np.uint32)).all()
    assert (n_inl == len(good)).any()
    for im in imgs:
        im.close()


def test_queries_without_any_sweep_record_give_no_match():
    """bench.py's ceiling measurement runs the pipeline on a sweep whose digest is
    switched off (timing build): every record then says "no group".  The
    neighbour extraction has to treat such a query as one without neighbours -- it
    must not turn the empty record into a row to gather."""
    try:
        tctx = matcher.Context(0, timing_build=True)
    except OSError:
        pytest.skip("no timing build of the library in this tree")
    rng = np.random.default_rng(77)
    images = []
    for v in range(4):
        n = 700 + 64 * v
        d = synth.sift_like(n, 900 + v)
        kp = np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
        images.append(matcher.Image(tctx, d, kp))
    pairs = np.array([[0, 1], [2, 3], [1, 2]], np.int32)
    kw = dict(ratio=0.7, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0,
              n_hyp=64)
    g_ref, i_ref = matcher.match_pairs(tctx, images, pairs, None, **kw)
    assert g_ref.sum() >= 0
    tctx.check(tctx._L.sfm_gpu_debug_set_knn_variant(tctx.handle, 4 + 16 * 7))   # MFMAs and fragment reads only
    g, i = matcher.match_pairs(tctx, images, pairs, None, **kw)
    assert (g == 0).all() and (i == 0).all()
    tctx.check(tctx._L.sfm_gpu_debug_set_knn_variant(tctx.handle, -1))
    g2, i2 = matcher.match_pairs(tctx, images, pairs, None, **kw)
    assert (g2 == g_ref).all() and (i2 == i_ref).all()
    for im in images:
        im.close()
    tctx.close()


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_images_prepared_in_device_memory_match_owned_images(gpu_ctx, dtype):
    """Descriptors that already sit in device memory (float rows as
    detectAndCompute leaves them, main.cpp:178, or bytes) converted by one
    batched call into caller-owned blocks: same pipeline results as images the
    library copied from the host, and the conversion can be repeated."""
    import torch
    n, n_img = 1300, 4
    rng = np.random.default_rng(77)
    pool = synth.sift_like(2000, 78)
    descs = [np.clip(pool[rng.permutation(2000)[:n]] + rng.integers(-4, 5, (n, 128)), 0, 255).astype(np.float32)
             for _ in range(n_img)]
    kps = [np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
           for _ in range(n_img)]
    pairs = np.array([[0, 1], [1, 2], [2, 3], [3, 0]], np.int32)
    kw = dict(ratio=0.8, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99,
              threshold=3.0, n_hyp=64)
    owned = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    g0, i0, s0 = matcher.match_pairs(gpu_ctx, owned, pairs, None, want_matches=True, **kw)

    d_desc = torch.from_numpy(np.stack(descs).astype(dtype)).cuda()
    d_kp = torch.from_numpy(np.stack(kps)).cuda()
    d_packed = torch.zeros((n_img, matcher.image_packed_bytes(n)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    prep = matcher.PrepareBatch(gpu_ctx, [d_desc[i].data_ptr() for i in range(n_img)], [n] * n_img,
                                128 * np.dtype(dtype).itemsize,
                                matcher.DESC_U8 if dtype == np.uint8 else matcher.DESC_F32,
                                [d_kp[i].data_ptr() for i in range(n_img)],
                                [d_packed[i].data_ptr() for i in range(n_img)])
    wrapped = [matcher.WrappedImage(gpu_ctx, d_packed[i].data_ptr(), n) for i in range(n_img)]
    for _ in range(2):
        prep.run()
        g1, i1, s1 = matcher.match_pairs(gpu_ctx, wrapped, pairs, None, want_matches=True, **kw)
        matcher.check_inputs(gpu_ctx)
        assert (g1 == g0).all() and (i1 == i0).all() and g0.sum() > 0
        for a, b in zip(s0, s1):
            assert (a == b).all()
    for im in owned + wrapped:
        im.close()


def test_batches_in_flight_return_what_the_blocking_call_returns(gpu_ctx):
    """begin/end with two batches on the stream at once, a blocking call queued
    between them, and the misuse errors."""
    from sfm_amd import _lib
    rng = np.random.default_rng(91)
    pool = synth.sift_like(2500, 92)
    images = []
    for v in range(5):
        n = 900 + 211 * v
        d = np.clip(pool[rng.permutation(2500)[:n]] + rng.integers(-4, 5, (n, 128)), 0, 255).astype(np.float32)
        kp = np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
        images.append(matcher.Image(gpu_ctx, d, kp))
    pa = np.array([[0, 1], [1, 2], [2, 3]], np.int32)
    pb = np.array([[4, 0], [3, 1]], np.int32)
    kw = dict(ratio=0.8, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99,
              threshold=3.0, n_hyp=96)
    want_a = matcher.match_pairs(gpu_ctx, images, pa, None, **kw)
    want_b = matcher.match_pairs(gpu_ctx, images, pb, None, **kw)
    assert want_a[0].sum() > 0 and want_b[0].sum() > 0
    a = matcher.PairBatch(gpu_ctx, images, pa, None, **kw)
    b = matcher.PairBatch(gpu_ctx, images, pb, None, **kw)
    with pytest.raises(_lib.SfmGpuError) as e:
        a.end()                                       # nothing in flight
    assert e.value.code == _lib.ERR_ARG
    for _ in range(3):
        a.begin()
        b.begin()
        with pytest.raises(_lib.SfmGpuError) as e:
            a.begin()                                 # a third batch
        assert e.value.code == _lib.ERR_CAPACITY
        mid = matcher.match_pairs(gpu_ctx, images, pb, None, **kw)   # queues behind both
        got_a = a.end()                               # oldest first
        got_b = b.end()
        for got, want in ((got_a, want_a), (got_b, want_b), (mid, want_b)):
            assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    # adaptive mode goes through the same entry points (begin then blocks)
    kwa = dict(ratio=0.8, model=matcher.MODEL_ESSENTIAL, prob=0.99, threshold=1.0)
    want = matcher.match_pairs(gpu_ctx, images, pa, synth.REF_K, **kwa)
    c = matcher.PairBatch(gpu_ctx, images, pa, synth.REF_K, **kwa)
    c.begin()
    got = c.end()
    assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    for im in images:
        im.close()


def test_conversion_on_a_second_context_ordered_behind_the_front_of_a_batch(gpu_ctx):
    """Two contexts on one GPU: descriptors of the next batch are converted on
    the second one's stream, which is told to wait for the front of the batch in
    flight on the first (sfm_gpu_stream_wait_front); results are those of the
    plain sequence."""
    import torch
    n, n_img = 1100, 4
    rng = np.random.default_rng(123)
    pool = synth.sift_like(1800, 124)
    sets = []
    for s in range(2):                                  # two different batches of images
        descs = [np.clip(pool[rng.permutation(1800)[:n]] + rng.integers(-4, 5, (n, 128)), 0, 255).astype(np.float32)
                 for _ in range(n_img)]
        kps = [np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32)
               for _ in range(n_img)]
        sets.append((descs, kps))
    pairs = np.array([[0, 1], [1, 2], [2, 3], [3, 0]], np.int32)
    kw = dict(ratio=0.8, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99,
              threshold=3.0, n_hyp=64)
    want = []
    for descs, kps in sets:
        owned = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
        want.append(matcher.match_pairs(gpu_ctx, owned, pairs, None, **kw))
        for im in owned:
            im.close()
    side = matcher.Context(0)
    matcher.stream_wait_front(side, gpu_ctx)            # nothing enqueued since: a no-op
    lanes = []
    for descs, kps in sets:
        d_desc = torch.from_numpy(np.stack(descs)).cuda()
        d_kp = torch.from_numpy(np.stack(kps)).cuda()
        d_packed = torch.zeros((n_img, matcher.image_packed_bytes(n)), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        prep = matcher.PrepareBatch(side, [d_desc[i].data_ptr() for i in range(n_img)], [n] * n_img, 512,
                                    matcher.DESC_F32, [d_kp[i].data_ptr() for i in range(n_img)],
                                    [d_packed[i].data_ptr() for i in range(n_img)])
        images = [matcher.WrappedImage(gpu_ctx, d_packed[i].data_ptr(), n) for i in range(n_img)]
        lanes.append((prep, matcher.PairBatch(gpu_ctx, images, pairs, None, **kw), images, d_desc, d_kp, d_packed))
    lanes[0][0].run()
    side.synchronize()
    for rep in range(4):
        cur, nxt = lanes[rep % 2], lanes[(rep + 1) % 2]
        cur[1].begin()
        matcher.stream_wait_front(side, gpu_ctx)        # the other set's conversion beside this batch's scoring
        nxt[0].run()
        got = cur[1].end()
        side.synchronize()
        assert (got[0] == want[rep % 2][0]).all() and (got[1] == want[rep % 2][1]).all()
    matcher.check_inputs(side)
    for lane in lanes:
        for im in lane[2]:
            im.close()
    side.close()


@pytest.mark.parametrize("mode", ["fixed", "adaptive"])
def test_begin_end_leaves_each_pairs_select_matches_on_device_and_host(gpu_ctx, oracle, mode):
    """The records form of begin/end: what match() returns per pair
    (main.cpp:225-231, 267) lands in caller-owned device memory and, behind the
    batch on a side stream, in a pinned host buffer -- bytes equal to the
    oracle pipeline's selectMatches and to the blocking call's, two batches in
    flight, the second one smaller and with other images, slots reused."""
    import torch
    from sfm_amd import _lib
    rng = np.random.default_rng(2718)
    descs, kps, images = [], [], []
    for v in range(5):
        n = 1000 + 173 * v
        descs.append(synth.sift_like(n, 2719 + v))
        kps.append(np.stack([rng.uniform(0, synth.IMG_W, n), rng.uniform(0, synth.IMG_H, n)], 1).astype(np.float32))
    # planted matches with a common geometry for the pairs used below: pair s uses rows
    # [200 s, 200 s + 100) of its query image and [200 s + 100, 200 s + 200) of its train
    # image, so that no pair overwrites what another planted
    for s_, (a, b) in enumerate([(0, 1), (1, 2), (2, 3), (4, 0), (3, 1)]):
        na, nb = len(descs[a]), len(descs[b])
        cnt = 100
        rq, rt = 200 * s_ + np.arange(cnt), 200 * s_ + 100 + np.arange(cnt)
        seed = s_ + 1
        descs[b][rt] = np.clip(descs[a][rq] + np.rint(rng.normal(0, 5, (cnt, 128))), 0, 255).astype(np.float32)
        planted = np.full(na, -1, np.int64)
        planted[rq] = rt
        kq, kt = synth.keypoints_for_pair(na, nb, planted, 60 + seed, wrong_frac=0.1)
        kps[a][rq] = kq[rq]
        kps[b][rt] = kt[rt]
    images = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    pa = np.array([[0, 1], [1, 2], [2, 3]], np.int32)
    pb = np.array([[4, 0], [3, 1]], np.int32)
    if mode == "fixed":
        K, model = None, matcher.MODEL_FUNDAMENTAL
        kw = dict(ratio=0.8, model=model, mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=96)
        okw = dict(prob=0.99, threshold=3.0, fixed_count=96)
    else:
        K, model = synth.REF_K, matcher.MODEL_ESSENTIAL
        kw = dict(ratio=0.8, model=model, prob=0.99, threshold=1.0)
        okw = dict(prob=0.99, threshold=1.0, max_iters=1000)
    cap = max(len(d) for d in descs)
    want = {}
    for name, pl in (("a", pa), ("b", pb)):
        g, i, sel = matcher.match_pairs(gpu_ctx, images, pl, K, want_matches=True, **kw)
        for p, (qi, ti) in enumerate(pl):
            ng, osel = _oracle_pair(oracle, descs[qi], descs[ti], kps[qi], kps[ti], K, 0.8, model, **okw)
            assert g[p] == ng and i[p] == len(osel) and sel[p].tobytes() == osel.tobytes()
        want[name] = (g, i, sel)
        assert i.min() > 40
    dev = {k: torch.full((len(pl), cap, 4), -7, dtype=torch.int32, device="cuda") for k, pl in (("a", pa), ("b", pb))}
    host = {k: torch.full((len(pl), cap, 4), -9, dtype=torch.int32).pin_memory() for k, pl in (("a", pa), ("b", pb))}
    torch.cuda.synchronize()
    ba = matcher.PairBatch(gpu_ctx, images, pa, K, records_dev=dev["a"].data_ptr(), match_capacity=cap,
                           records_host=host["a"].data_ptr(), **kw)
    bb = matcher.PairBatch(gpu_ctx, images, pb, K, records_dev=dev["b"].data_ptr(), match_capacity=cap,
                           records_host=host["b"].data_ptr(), **kw)
    only_dev = matcher.PairBatch(gpu_ctx, images, pb, K, records_dev=dev["b"].data_ptr(), match_capacity=cap, **kw)
    for rep in range(3):
        ba.begin()
        bb.begin()
        got = {"a": ba.end(), "b": bb.end()}
        gpu_ctx.synchronize()
        for name in ("a", "b"):
            g, i, sel = want[name]
            assert (got[name][0] == g).all() and (got[name][1] == i).all()
            h = host[name].numpy().view(_lib.DMATCH_DTYPE).reshape(len(g), cap)
            d = dev[name].cpu().numpy().view(_lib.DMATCH_DTYPE).reshape(len(g), cap)
            for p in range(len(g)):
                assert h[p, :i[p]].tobytes() == sel[p].tobytes()
                assert d[p, :i[p]].tobytes() == sel[p].tobytes()
                # beyond a pair's records the host buffer is unspecified: untouched, or the bytes of the
                # device buffer (the batch's whole record area may travel as one copy)
                rest = host[name][p, i[p]:]
                assert ((rest == -9) | (rest == -7)).all()
            host[name].fill_(-9)
            dev[name].fill_(-7)
        torch.cuda.synchronize()
    # device records without a host copy
    only_dev.begin()
    g, i = only_dev.end()
    d = dev["b"].cpu().numpy().view(_lib.DMATCH_DTYPE).reshape(len(pb), cap)
    for p in range(len(pb)):
        assert d[p, :i[p]].tobytes() == want["b"][2][p].tobytes()
    # misuse: pageable host memory, capacity below the largest query image
    pageable = np.zeros((len(pb), cap), _lib.DMATCH_DTYPE)
    bad = matcher.PairBatch(gpu_ctx, images, pb, K, records_dev=dev["b"].data_ptr(), match_capacity=cap,
                            records_host=pageable.ctypes.data, **kw)
    with pytest.raises(_lib.SfmGpuError) as e:
        bad.begin()
    assert e.value.code == _lib.ERR_ARG
    small = matcher.PairBatch(gpu_ctx, images, pb, K, records_dev=dev["b"].data_ptr(), match_capacity=100, **kw)
    with pytest.raises(_lib.SfmGpuError) as e:
        small.begin()
    assert e.value.code == _lib.ERR_CAPACITY
    for im in images:
        im.close()


def test_adaptive_groups_in_flight_with_pairs_that_want_more_sample_rows(gpu_ctx, oracle):
    """sfm_gpu_match_pairs keeps two launch groups on the stream and enqueues the
    adaptive loop's first batch of sample rows without waiting; pairs whose loop
    wants more (low inlier share: hundreds of iterations) make their group run
    again.  261 pairs = two groups, a mix of overlapping pairs (done in the
    first batch), pairs with about half their matches wrong (more batches) and
    pairs with nothing in common: counts and records equal one-pair calls, and
    the oracle pipeline on a sample."""
    rng = np.random.default_rng(77)
    n_img = 14
    descs = [synth.sift_like(700 + 31 * v, 900 + v) for v in range(n_img)]
    kps = [np.stack([rng.uniform(0, synth.IMG_W, len(d)), rng.uniform(0, synth.IMG_H, len(d))], 1).astype(np.float32)
           for d in descs]
    # image v+1 repeats 200 rows of image v; half of those keep a random keypoint (wrong geometry)
    for v in range(n_img - 1):
        rq, rt = 40 * (v % 3) + np.arange(200), 300 + np.arange(200)
        descs[v + 1][rt] = np.clip(descs[v][rq] + np.rint(rng.normal(0, 5, (200, 128))), 0, 255).astype(np.float32)
        planted = np.full(len(descs[v]), -1, np.int64)
        planted[rq] = rt
        kq, kt = synth.keypoints_for_pair(len(descs[v]), len(descs[v + 1]), planted, 40 + v,
                                          wrong_frac=0.5 if v % 2 else 0.05)
        kps[v][rq] = kq[rq]
        kps[v + 1][rt] = kt[rt]
    images = [matcher.Image(gpu_ctx, d, k) for d, k in zip(descs, kps)]
    all_pairs = [[i, j] for i in range(n_img) for j in range(n_img) if i != j]
    pairs = np.array((all_pairs * 2)[:261], np.int32)
    kw = dict(ratio=0.8, model=matcher.MODEL_ESSENTIAL, mode=matcher.RANSAC_ADAPTIVE, prob=0.99, threshold=1.0,
              max_iters=400)
    ng, ni, sel = matcher.match_pairs(gpu_ctx, images, pairs, synth.REF_K, want_matches=True, **kw)
    ng2, ni2 = matcher.match_pairs(gpu_ctx, images, pairs, synth.REF_K, **kw)          # counts only
    assert (ng2 == ng).all() and (ni2 == ni).all()
    checked = 0
    for p in list(range(0, 261, 9)) + [255, 256, 260]:
        i, j = pairs[p]
        g1, i1, s1 = matcher.match_pairs(gpu_ctx, images, pairs[p:p + 1], synth.REF_K, want_matches=True, **kw)
        assert g1[0] == ng[p] and i1[0] == ni[p] and s1[0].tobytes() == sel[p].tobytes()
        if j == i + 1:
            og, osel = _oracle_pair(oracle, descs[i], descs[j], kps[i], kps[j], synth.REF_K, 0.8, 0, prob=0.99,
                                    threshold=1.0, max_iters=400)
            assert og == ng[p] and osel.tobytes() == sel[p].tobytes()
            checked += 1
    assert checked >= 2 and (ni > 60).sum() >= 13
    # the same through begin / end with two batches in flight, the older one needing the second pass
    pa, pb = pairs[:40], pairs[40:70]
    cap = max(len(d) for d in descs)
    import torch
    from sfm_amd import _lib
    dev = [torch.full((len(pl_), cap, 4), -7, dtype=torch.int32, device="cuda") for pl_ in (pa, pb)]
    host = [torch.full((len(pl_), cap, 4), -9, dtype=torch.int32).pin_memory() for pl_ in (pa, pb)]
    torch.cuda.synchronize()
    ba = matcher.PairBatch(gpu_ctx, images, pa, synth.REF_K, records_dev=dev[0].data_ptr(), match_capacity=cap,
                           records_host=host[0].data_ptr(), **kw)
    bb = matcher.PairBatch(gpu_ctx, images, pb, synth.REF_K, records_dev=dev[1].data_ptr(), match_capacity=cap,
                           records_host=host[1].data_ptr(), **kw)
    for rep in range(2):
        ba.begin()
        bb.begin()
        for b_, lo, hi, h, d in ((ba, 0, 40, host[0], dev[0]), (bb, 40, 70, host[1], dev[1])):
            g, c = b_.end()
            assert (g == ng[lo:hi]).all() and (c == ni[lo:hi]).all()
            hh = h.numpy().view(_lib.DMATCH_DTYPE).reshape(hi - lo, cap)
            dd = d.cpu().numpy().view(_lib.DMATCH_DTYPE).reshape(hi - lo, cap)
            for p in range(hi - lo):
                assert hh[p, :c[p]].tobytes() == sel[lo + p].tobytes() == dd[p, :c[p]].tobytes()
    for im in images:
        im.close()
