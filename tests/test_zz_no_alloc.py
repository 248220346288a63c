"""The entry point that allocates nothing (sfm_gpu_knn_match_images_noalloc)
against sfm_gpu_knn_match_images on the same prepared images, through the C ABI.

Bar: every output byte equal (cv::DMatch records incl. float distance bits, and
the neighbour counts), on two calls in a row, the second on a smaller input --
the static working memory keeps the first call's contents.  The existing entry
point is itself held to the CPU oracle in test_gpu_knn.py; the first case here
is checked against the oracle too.
"""
import ctypes

import numpy as np
import pytest

from sfm_amd import _lib, matcher, synth

pytestmark = pytest.mark.gpu


def _both(ctx, qi, ti, k):
    import torch
    nq = qi.n
    L = ctx._L
    out = []
    for fn in ("plain", "noalloc"):
        # poisoned outputs: whatever a call does not write shows up
        m = torch.full((max(nq, 1) * k * 4,), -7, dtype=torch.int32, device="cuda")
        c = torch.full((max(nq, 1),), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        if fn == "plain":
            ctx.check(L.sfm_gpu_knn_match_images(ctx.handle, qi.handle, ti.handle, k,
                                                 ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(c.data_ptr()), 1))
        else:
            ctx.check(L.sfm_gpu_knn_match_images_noalloc(ctx.handle, qi.handle, ti.handle, k,
                                                         ctypes.c_void_p(m.data_ptr()),
                                                         ctypes.c_void_p(c.data_ptr())))
        ctx.synchronize()
        out.append((m.cpu().numpy()[: nq * k * 4].view(_lib.DMATCH_DTYPE).reshape(nq, k), c.cpu().numpy()[:nq]))
    return out


def _check_pair(ctx, q, t, k, oracle=None):
    qi, ti = matcher.Image(ctx, q), matcher.Image(ctx, t)
    (pm, pc), (nm, nc) = _both(ctx, qi, ti, k)
    assert (pc == nc).all()
    assert pm.tobytes() == nm.tobytes()
    if oracle is not None:
        om, oc = oracle.knn_match(q, t, k, nthreads=8)
        assert (nc == oc).all()
        assert (nm["trainIdx"] == om["trainIdx"]).all()
        assert (nm["distance"].view(np.uint32) == om["distance"].view(np.uint32)).all()
    qi.close()
    ti.close()


def test_noalloc_matches_existing_entry_point_twice_second_smaller(gpu_ctx, oracle):
    # first call: several train parts, ties and rows for the float-ranked pass
    q1, t1, _ = synth.planted_pair(3000, 9000, 11, 12, 13)
    t1[100:140] = t1[60:100]          # exact duplicates: index tie-break + redo list in use
    _check_pair(gpu_ctx, q1, t1, 2, oracle)
    # second call right behind it on a smaller input: stale records, a stale
    # redo list and a stale job descriptor from the first call are all in reach
    q2, t2, _ = synth.planted_pair(700, 1300, 21, 22, 23)
    _check_pair(gpu_ctx, q2, t2, 2, oracle)
    # and k = 1, a single train row, far colliding distances
    _check_pair(gpu_ctx, q2[:33], t2[:1], 1)
    far = np.zeros((64, 128), np.float32)
    far_t = np.full((300, 128), 255, np.float32)
    far_t[::2, 0] = 254
    _check_pair(gpu_ctx, far, far_t, 2, oracle)


def test_noalloc_rejects_sizes_beyond_its_static_memory(gpu_ctx):
    import torch
    q = synth.uniform_u8(32769, 5)
    t = synth.uniform_u8(64, 6)
    qi, ti = matcher.Image(gpu_ctx, q), matcher.Image(gpu_ctx, t)
    m = torch.zeros(32769 * 2 * 4, dtype=torch.int32, device="cuda")
    c = torch.zeros(32769, dtype=torch.int32, device="cuda")
    rc = gpu_ctx._L.sfm_gpu_knn_match_images_noalloc(gpu_ctx.handle, qi.handle, ti.handle, 2,
                                                     ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(c.data_ptr()))
    assert rc == _lib.ERR_CAPACITY
    assert b"static working memory" in gpu_ctx._L.sfm_gpu_last_error_string(gpu_ctx.handle)
    qi.close()
    ti.close()


def test_noalloc_allocates_no_device_memory_on_a_fresh_context():
    """A context that has never run anything (all its grow-only workspaces at zero
    capacity) goes straight to the entry point: free device memory is the same
    before and after, and the results are those of the allocating entry point
    on another context."""
    import torch
    q, t, _ = synth.planted_pair(2100, 5000, 31, 32, 33)
    other = matcher.Context(0)
    qo, to = matcher.Image(other, q), matcher.Image(other, t)
    want_m, want_c = matcher.knn_match_images(other, qo, to, 2)
    fresh = matcher.Context(0)
    qi, ti = matcher.Image(fresh, q), matcher.Image(fresh, t)   # images allocate; the call under test must not
    m = torch.full((2100 * 2 * 4,), -7, dtype=torch.int32, device="cuda")
    c = torch.full((2100,), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    fresh.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(2):
        fresh.check(fresh._L.sfm_gpu_knn_match_images_noalloc(fresh.handle, qi.handle, ti.handle, 2,
                                                              ctypes.c_void_p(m.data_ptr()),
                                                              ctypes.c_void_p(c.data_ptr())))
        fresh.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0, "device memory changed by %d bytes" % (free0 - free1)
    got_m = m.cpu().numpy().view(_lib.DMATCH_DTYPE).reshape(2100, 2)
    assert got_m.tobytes() == want_m.tobytes() and (c.cpu().numpy() == want_c).all()
    for im in (qi, ti, qo, to):
        im.close()
    fresh.close()
    other.close()
