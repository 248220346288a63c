"""GPU parity of the k-NN path against the CPU oracle, through the C ABI.

Bar: bit-exact trainIdx and float32 distance bits for every query row.
"""
import numpy as np
import pytest

from sfm_amd import matcher, synth

pytestmark = pytest.mark.gpu


def _assert_same(gm, gc, om, oc):
    assert (gc == oc).all()
    assert (gm["queryIdx"] == om["queryIdx"]).all()
    assert (gm["imgIdx"] == 0).all()
    bad = np.nonzero(gm["trainIdx"] != om["trainIdx"])
    assert len(bad[0]) == 0, (bad[0][:10], gm[bad][:5], om[bad][:5])
    valid = om["trainIdx"] >= 0
    assert (gm["distance"].view(np.uint32)[valid] == om["distance"].view(np.uint32)[valid]).all()


def test_mfma_operand_and_accumulator_layout(gpu_ctx):
    import ctypes
    rng = np.random.default_rng(0)
    a = rng.integers(-128, 128, (32, 32)).astype(np.int8)
    b = rng.integers(-128, 128, (32, 32)).astype(np.int8)   # [col][k]
    c = np.zeros((32, 32), np.int32)
    L = gpu_ctx._L
    gpu_ctx.check(L.sfm_gpu_debug_mfma_probe(gpu_ctx.handle, matcher.ptr(a), matcher.ptr(b),
                                             matcher.ptr(c)))
    ref = a.astype(np.int32) @ b.astype(np.int32).T
    assert (c == ref).all()


def test_device_sqrtf_is_correctly_rounded_on_all_squared_distances(gpu_ctx):
    # the match records take their float distance from the device's sqrtf; this
    # pins that it equals the correctly rounded root for every integer < 2^24
    m = np.zeros(1, np.int32)
    gpu_ctx.check(gpu_ctx._L.sfm_gpu_debug_sqrt_probe(gpu_ctx.handle, matcher.ptr(m)))
    assert m[0] == 0


def test_mfma16_operand_and_accumulator_layout(gpu_ctx):
    rng = np.random.default_rng(1)
    a = rng.integers(-128, 128, (16, 64)).astype(np.int8)
    b = rng.integers(-128, 128, (16, 64)).astype(np.int8)   # [col][k]
    c = np.zeros((16, 16), np.int32)
    gpu_ctx.check(gpu_ctx._L.sfm_gpu_debug_mfma16_probe(gpu_ctx.handle, matcher.ptr(a), matcher.ptr(b),
                                                         matcher.ptr(c)))
    assert (c == a.astype(np.int32) @ b.astype(np.int32).T).all()


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("nq,nt", [(1, 1), (5, 2), (33, 257), (257, 1000), (700, 3000)])
def test_knn_small_shapes(gpu_ctx, oracle, variant, nq, nt):
    gpu_ctx._L.sfm_gpu_debug_set_knn_variant(gpu_ctx.handle, variant)
    q = synth.uniform_u8(nq, 100 + nq)
    t = synth.uniform_u8(nt, 200 + nt)
    gm, gc = matcher.BFMatcher(gpu_ctx).knnMatch(q, t, 2)
    om, oc = oracle.knn_match(q, t, 2, nthreads=8)
    _assert_same(gm, gc, om, oc)
    gpu_ctx._L.sfm_gpu_debug_set_knn_variant(gpu_ctx.handle, -1)


@pytest.fixture(params=[-1, 0, 5], ids=["planner_mfma16x16", "mfma32x32", "mfma16x16_8waves"])
def sweep_shape(request, gpu_ctx):
    """Runs a test with the planner's choice and with each MFMA shape of the sweep forced."""
    gpu_ctx._L.sfm_gpu_debug_set_knn_variant(gpu_ctx.handle, request.param)
    yield request.param
    gpu_ctx._L.sfm_gpu_debug_set_knn_variant(gpu_ctx.handle, -1)


def test_knn_sift_like_planted(gpu_ctx, oracle, sweep_shape):
    Q, T, planted = synth.planted_pair(4096, 4096, 1001, 1002, 1003)
    gm, gc = matcher.BFMatcher(gpu_ctx).knnMatch(Q, T, 2)
    om, oc = oracle.knn_match(Q, T, 2, nthreads=8)
    _assert_same(gm, gc, om, oc)
    keep = matcher.ratio_test(gm, gc, 0.7)
    assert keep.tolist() == oracle.ratio_test(om, oc, 0.7).tolist()
    assert (gm["trainIdx"][keep, 0] == planted[keep]).mean() > 0.99


def test_knn_ties_across_tiles_chunks_and_splits(gpu_ctx, oracle, sweep_shape):
    # duplicated train rows far apart: ties must resolve to the lowest row
    rng = np.random.default_rng(9)
    t = rng.integers(0, 256, (5000, 128)).astype(np.float32)
    q = rng.integers(0, 256, (64, 128)).astype(np.float32)
    for i, (a, b, c) in enumerate([(3, 35, 4000), (0, 255, 256), (511, 512, 4999), (100, 131, 163),
                                   (1023, 1024, 2048), (31, 32, 33), (4095, 4096, 4097)]):
        q[i] = rng.integers(0, 256, 128)
        t[a] = q[i]
        t[b] = q[i]
        t[c] = q[i]
        q[i, 0] = (q[i, 0] + 1) % 256    # all three copies at the same non-zero distance
    # exact duplicate train rows at zero distance as well
    t[700] = q[40]
    t[2900] = q[40]
    # odd/even Tn rows with equal d2: same key high bits, parity differs
    gm, gc = matcher.BFMatcher(gpu_ctx).knnMatch(q, t, 2)
    om, oc = oracle.knn_match(q, t, 2, nthreads=8)
    _assert_same(gm, gc, om, oc)
    assert gm["trainIdx"][0].tolist() == [3, 35]
    assert gm["trainIdx"][40].tolist() == [700, 2900]


def test_knn_extremes_and_float_collision(gpu_ctx, oracle, sweep_shape):
    # all-0 / all-255 rows (d2 = 8323200) and engineered d2 >= 2^22 collisions
    q = np.zeros((4, 128), np.float32)
    q[1] = 255
    base = None
    for d2 in range(1 << 22, (1 << 22) + 4000):
        if np.float32(np.sqrt(np.float64(d2))) == np.float32(np.sqrt(np.float64(d2 + 1))):
            base = d2
            break

    def row_with_d2(target):
        r = np.zeros(128, np.float32)
        rem = target
        for i in range(128):
            v = min(255, int(np.floor(np.sqrt(rem))))
            r[i] = v
            rem -= v * v
        assert rem == 0
        return r
    t = np.stack([row_with_d2(base + 1), row_with_d2(base), np.full(128, 255, np.float32),
                  np.zeros(128, np.float32)] + [np.full(128, 255, np.float32)] * 300)
    t[3] = 255   # no exact zero-distance row for query 0
    gm, gc = matcher.BFMatcher(gpu_ctx).knnMatch(q, t, 2)
    om, oc = oracle.knn_match(q, t, 2)
    _assert_same(gm, gc, om, oc)
    assert gm["trainIdx"][0].tolist() == [0, 1]


def test_knn_k1_u8_strided_and_empty(gpu_ctx, oracle):
    rng = np.random.default_rng(4)
    big = rng.integers(0, 256, (300, 160)).astype(np.float32)
    q = big[:, :128]            # row stride 640 bytes
    t8 = rng.integers(0, 256, (900, 128)).astype(np.uint8)
    t = t8.astype(np.float32)
    m = matcher.BFMatcher(gpu_ctx)
    gm, gc = m.knnMatch(q, t, 1)
    om, oc = oracle.knn_match(np.ascontiguousarray(q), t, 1, nthreads=4)
    _assert_same(gm, gc, om, oc)
    gm2, gc2 = m.knnMatch(np.ascontiguousarray(q).astype(np.uint8), t8, 2)
    om2, oc2 = oracle.knn_match(np.ascontiguousarray(q), t, 2, nthreads=4)
    _assert_same(gm2, gc2, om2, oc2)
    gm3, gc3 = m.knnMatch(q, np.zeros((0, 128), np.float32), 2)
    assert (gc3 == 0).all() and (gm3["trainIdx"] == -1).all()
    gm4, gc4 = m.knnMatch(np.zeros((0, 128), np.float32), t, 2)
    assert gm4.shape == (0, 2)


def test_knn_rejects_non_integer_values(gpu_ctx):
    q = synth.uniform_u8(10, 1)
    t = synth.uniform_u8(10, 2)
    t[3, 5] = 12.5
    with pytest.raises(matcher.SfmGpuError) as e:
        matcher.BFMatcher(gpu_ctx).knnMatch(q, t, 2)
    assert e.value.code == -3
    t[3, 5] = 256.0
    with pytest.raises(matcher.SfmGpuError):
        matcher.BFMatcher(gpu_ctx).knnMatch(q, t, 2)
    with pytest.raises(matcher.SfmGpuError) as e:
        matcher.BFMatcher(gpu_ctx).knnMatch(q, synth.uniform_u8(10, 2), 3)
    assert e.value.code == -5


def test_knn_full_size_8k_against_oracle(gpu_ctx, oracle, sweep_shape):
    Q, T, _ = synth.planted_pair(8192, 8192, 1001, 1002, 1003)
    gm, gc = matcher.BFMatcher(gpu_ctx).knnMatch(Q, T, 2)
    om, oc = oracle.knn_match(Q, T, 2, nthreads=16)
    _assert_same(gm, gc, om, oc)


def test_knn_32k_self_match_property(gpu_ctx):
    # size-independent property at the full 32768 x 32768 size: matching a set
    # against a row-permuted copy of itself finds the permutation at distance 0,
    # and the second neighbour is never closer than the first
    Q = synth.sift_like(32768, 1101)
    perm = np.random.default_rng(1103).permutation(32768)
    T = Q[perm]
    gm, gc = matcher.BFMatcher(gpu_ctx).knnMatch(Q, T, 2)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(32768)
    # duplicates in Q (if any) would tie at distance 0 and resolve to the lower row
    assert (gm["distance"][:, 0] == 0).all()
    same = gm["trainIdx"][:, 0] == inv
    assert same.mean() > 0.999
    assert (gm["distance"][:, 1] >= gm["distance"][:, 0]).all()
    assert (gc == 2).all()


def _numpy_top2(Q, T, rows):
    """Exact top-2 of the given query rows by integer arithmetic in numpy:
    (trainIdx[2], float32 distance[2]) with OpenCV's order (float distance, then
    ascending train index)."""
    Ti = T.astype(np.int64)
    out_idx, out_d = [], []
    for r in rows:
        d2 = ((Ti - Q[r].astype(np.int64)) ** 2).sum(1)
        d = np.sqrt(d2.astype(np.float64)).astype(np.float32)   # d2 < 2^24: sqrt correctly rounded via double
        order = np.lexsort((np.arange(len(T)), d))[:2]
        out_idx.append(order)
        out_d.append(d[order])
    return np.array(out_idx), np.array(out_d)


@pytest.mark.parametrize("nq,nt,seed", [(20000, 20000, 1201), (32768, 32768, 1202), (9001, 30011, 1203)])
def test_knn_large_ragged_sizes_against_numpy_sample(gpu_ctx, nq, nt, seed, sweep_shape):
    # sizes of the larger configurations (20k x 20k views, 32k x 32k pair, a ragged
    # pair): a random sample of query rows is checked against an independent exact
    # computation, which does not depend on the size the CPU oracle can finish
    Q, T, _ = synth.planted_pair(nq, nt, seed, seed + 1, seed + 2)
    gm, gc = matcher.BFMatcher(gpu_ctx).knnMatch(Q, T, 2)
    assert (gc == 2).all() and (gm["queryIdx"][:, 0] == np.arange(nq)).all()
    rows = np.random.default_rng(seed).choice(nq, 96, replace=False)
    idx, dist = _numpy_top2(Q, T, rows)
    assert (gm["trainIdx"][rows] == idx).all()
    assert (gm["distance"][rows].view(np.uint32) == dist.view(np.uint32)).all()
    assert (gm["distance"][:, 1] >= gm["distance"][:, 0]).all()


def test_knn_argument_errors_and_limits(gpu_ctx):
    from sfm_amd import _lib
    bf = matcher.BFMatcher(gpu_ctx)
    q = np.zeros((4, 128), np.float32)
    with pytest.raises(_lib.SfmGpuError) as e:
        bf.knnMatch(q, q, 3)                       # k > 2 is not on the path
    assert e.value.code == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.SfmGpuError) as e:
        bf.knnMatch(np.zeros((4, 64), np.float32), np.zeros((4, 64), np.float32), 2)   # dim != 128
    assert e.value.code == _lib.ERR_ARG
    with pytest.raises(_lib.SfmGpuError) as e:       # cv::BFMatcher's own limit on the train set
        bf.knnMatch(np.zeros((1, 128), np.uint8), np.zeros((1 << 18, 128), np.uint8), 2)
    assert e.value.code == _lib.ERR_TOO_MANY
    # the largest train set that is allowed: distances of an all-zero query to rows
    # that are zero except one marked row
    T = np.zeros(((1 << 18) - 1, 128), np.uint8)
    T[:, 0] = 9
    T[200001, 0] = 1
    T[77, 0] = 2
    gm, gc = bf.knnMatch(np.zeros((3, 128), np.uint8), T, 2)
    assert (gm["trainIdx"] == [200001, 77]).all() and (gm["distance"] == [1.0, 2.0]).all()
