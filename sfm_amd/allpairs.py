"""Exhaustive (or ring) pair matching over several GPUs, one process per GPU.

The reference matches image i against image i+1 in a sequential loop
(/root/reference/main.cpp:167-173).  Pairs are independent, so the loop is
sharded: images are block-owned by rank, prepared once where they live, their
prepared blocks are exchanged with ONE all-gather (RCCL over xGMI on GPUs), the
pair list is dealt over the ranks, and the V x V result matrices are summed
with one all-reduce.  Nothing else crosses ranks.

The arithmetic is behind a small backend interface so that the exchange and
the dealing can be exercised on CPU with the gloo backend:

    backend.prepare(local_images) -> (blobs [n_local, blob_bytes] uint8 tensor,
                                      rows  [n_local] int32 tensor)
    backend.blob_bytes(max_rows)  -> int
    backend.match(blobs [V, blob_bytes], rows [V], pairs [P, 2])
                                  -> (n_good [P], n_inliers [P]) int arrays

`GpuBackend` is the product path (libsfm_gpu.so); there is no CPU backend in
this package.
"""
import numpy as np


def pair_list(n_images, mode="all"):
    """Pairs (i, j) to match.  "ring": i -> (i+1) mod V, the reference's loop
    (main.cpp:167-173); "all": every i < j."""
    V = int(n_images)
    if mode == "ring":
        if V < 2:
            return np.zeros((0, 2), np.int32)
        return np.array([[i, (i + 1) % V] for i in range(V)], np.int32)
    if mode == "all":
        i, j = np.triu_indices(V, 1)      # row-major: (0,1) (0,2) ... (1,2) ...
        return np.stack([i, j], 1).astype(np.int32).reshape(-1, 2)
    raise ValueError("unknown pair mode %r" % (mode,))


def spanning_order(counts):
    """Order in which to chain the images, from the V x V match-count (or
    inlier-count) matrix of an exhaustive run: the maximum spanning tree of the
    graph whose edge weights are those counts — what /root/reference/README.md:39
    names as the usual way to recover the capture order (the reference itself
    reads the files in name order and leaves this out, README.md:41).

    counts[i, j] and counts[j, i] are added (the matcher fills one of them, or
    both when both directions were matched).  Prim's algorithm, grown from the
    image with the largest total count; among equal weights the edge from the
    earliest-visited image wins, then the lower image number, so the result is
    deterministic.  An image no positive-count edge reaches starts a new tree.

    Returns (order [V], parent [V]): images in visiting order, and for each
    image the image it was attached to (-1 for the root of a tree)."""
    c = np.asarray(counts, dtype=np.int64)
    if c.ndim != 2 or c.shape[0] != c.shape[1]:
        raise ValueError("counts must be a square matrix")
    V = c.shape[0]
    w = c + c.T
    np.fill_diagonal(w, 0)
    order = np.zeros(V, np.int32)
    parent = np.full(V, -1, np.int32)
    done = np.zeros(V, bool)
    best_w = np.zeros(V, np.int64)        # heaviest edge from the tree to each outside image
    best_from = np.full(V, -1, np.int32)
    total = w.sum(1)
    n = 0
    while n < V:
        cand = np.where(~done, best_w, -1)
        v = int(np.argmax(cand))          # lowest index among equals
        if cand[v] <= 0:                  # nothing reachable: root of a new tree
            v = int(np.argmax(np.where(~done, total, -1)))
            parent[v] = -1
        else:
            parent[v] = best_from[v]
        done[v] = True
        order[n] = v
        n += 1
        better = ~done & (w[v] > best_w)  # strict: earlier-visited images keep their claim
        best_w[better] = w[v][better]
        best_from[better] = v
    return order, parent


def owned_range(n_images, world, rank):
    """Contiguous block of images a rank owns (sizes differ by at most one)."""
    base, extra = divmod(int(n_images), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def deal_pairs(pairs, rows, world):
    """Assigns every pair to a rank: pairs sorted by descending cost
    (rows_i * rows_j, ties by pair index) are dealt out in rounds, one pair per
    rank and round, the ranks taken in the order of their load so far (the
    least loaded rank gets the round's most expensive pair).  Loads then differ
    by at most the cost of one pair.  Deterministic, identical on every rank,
    and one vector operation per round, not per pair (499 500 pairs for 1000
    images).  Returns a list of index arrays, one per rank, each ascending."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    rows = np.asarray(rows, dtype=np.int64)
    world = int(world)
    cost = rows[pairs[:, 0]] * rows[pairs[:, 1]] if len(pairs) else np.zeros(0, np.int64)
    order = np.lexsort((np.arange(len(pairs)), -cost))
    owner = np.zeros(len(pairs), np.int64)
    if len(pairs) and (cost == cost[0]).all():
        # every pair costs the same (images of one size): the rounds leave the loads
        # level, so the rank order never changes
        owner[order] = np.arange(len(pairs)) % world
    else:
        load = np.zeros(world, np.int64)
        for r0 in range(0, len(order), world):
            idx = order[r0:r0 + world]
            ranks = np.lexsort((np.arange(world), load))[:len(idx)]   # least loaded first, ties by rank
            owner[idx] = ranks
            load[ranks] += cost[idx] + 1
    return [np.flatnonzero(owner == r).astype(np.int64) for r in range(world)]


def match_all_pairs(local_images, n_images, backend, mode="all", group=None, device=None, return_stats=False):
    """Matches all pairs of `n_images` images spread over the ranks of `group`.

    local_images: this rank's images, in global order (owned_range()).
    Returns (pairs [P,2], n_good [V,V], n_inliers [V,V]) on every rank; the
    matrices hold pair (i, j) at [i, j] and zeros elsewhere.  return_stats adds a
    dict with the bytes the all-gather delivered to this rank and its pair share.
    """
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    V = int(n_images)
    lo, hi = owned_range(V, world, rank)
    if len(local_images) != hi - lo:
        raise ValueError("rank %d owns images [%d, %d) but was given %d" % (rank, lo, hi, len(local_images)))

    # 1. agree on the block size (largest image anywhere)
    local_max = max([backend.rows_of(im) for im in local_images] + [0])
    t = torch.tensor([local_max], dtype=torch.int64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    max_rows = int(t.item())
    blob_bytes = backend.blob_bytes(max_rows)

    # 2. prepare where the images live
    blobs, rows = backend.prepare(local_images, max_rows)
    per_rank = (V + world - 1) // world  # owned_range gives at most this many
    stats = {"world": world, "blob_bytes": int(blob_bytes), "gathered_bytes": 0}
    if distributed and world > 1:
        # 3. one all-gather of equally sized slabs (short ranks pad with zeros).
        # RCCL moves device memory; a gloo group (CPU rehearsal of the same control
        # flow) is handed host copies.
        on_host = dist.get_backend(group) == "gloo" and blobs.is_cuda
        cdev = torch.device("cpu") if on_host else blobs.device
        slab = torch.zeros((per_rank, blob_bytes), dtype=torch.uint8, device=cdev)
        slab_rows = torch.zeros(per_rank, dtype=torch.int32, device=cdev)
        slab[: hi - lo] = blobs.to(cdev)
        slab_rows[: hi - lo] = rows.to(cdev)
        all_slabs = torch.empty((world * per_rank, blob_bytes), dtype=torch.uint8, device=cdev)
        all_rows = torch.empty(world * per_rank, dtype=torch.int32, device=cdev)
        dist.all_gather_into_tensor(all_slabs, slab, group=group)
        dist.all_gather_into_tensor(all_rows, slab_rows, group=group)
        stats["gathered_bytes"] = int(all_slabs.numel() + all_rows.numel() * 4)
        index = []
        for r in range(world):
            rlo, rhi = owned_range(V, world, r)
            index += [r * per_rank + k for k in range(rhi - rlo)]
        index = torch.tensor(index, dtype=torch.int64, device=cdev)
        all_blobs = all_slabs.index_select(0, index) if len(index) != world * per_rank else all_slabs
        all_rows = all_rows.index_select(0, index)
        if on_host:
            all_blobs = all_blobs.to(blobs.device)
    else:
        all_blobs, all_rows = blobs, rows

    # 4. deal the pairs, run this rank's share
    pairs = pair_list(V, mode)
    rows_host = all_rows.cpu().numpy()
    mine = deal_pairs(pairs, rows_host, world)[rank]
    good = torch.zeros((V, V), dtype=torch.int64, device=device)
    inl = torch.zeros((V, V), dtype=torch.int64, device=device)
    if len(mine):
        ng, ni = backend.match(all_blobs, rows_host, pairs[mine])
        pi = torch.as_tensor(pairs[mine, 0].astype(np.int64), device=device)
        pj = torch.as_tensor(pairs[mine, 1].astype(np.int64), device=device)
        good[pi, pj] = torch.as_tensor(np.asarray(ng, dtype=np.int64), device=device)
        inl[pi, pj] = torch.as_tensor(np.asarray(ni, dtype=np.int64), device=device)

    # 5. one all-reduce of the result matrices
    if distributed and world > 1:
        both = torch.stack([good, inl])
        if dist.get_backend(group) == "gloo" and both.is_cuda:
            both = both.cpu()
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        good, inl = both[0], both[1]
    if return_stats:
        stats["pairs_mine"] = int(len(mine))
        return pairs, good.cpu().numpy(), inl.cpu().numpy(), stats
    return pairs, good.cpu().numpy(), inl.cpu().numpy()


class GpuBackend:
    """libsfm_gpu.so on the current GPU.  An image is (descriptors [n,128]
    float32/uint8 numpy array or CUDA tensor, keypoints [n,2] float32, same).

    The library launches on the context's stream; tensors handed to it (CUDA
    descriptors, the zero-filled blocks, the all-gather's output) are produced on
    torch's current stream.  When the two differ the backend makes torch's stream
    wait before the library reads and the library's before torch reads; a context
    created on torch's stream (Context(dev, stream=torch.cuda.current_stream()
    .cuda_stream)) needs neither."""

    def __init__(self, ctx, **match_kwargs):
        self.ctx = ctx
        self.match_kwargs = match_kwargs

    def _torch_to_library(self):
        import torch
        cur = torch.cuda.current_stream()
        if self.ctx.stream is None or self.ctx.stream != cur.cuda_stream:
            cur.synchronize()  # what torch queued is done before the library's stream touches it

    @staticmethod
    def rows_of(image):
        return int(image[0].shape[0])

    @staticmethod
    def blob_bytes(max_rows):
        from . import matcher
        return matcher.image_packed_bytes(max_rows)

    def prepare(self, images, max_rows):
        import torch
        from . import matcher
        blk = matcher.image_packed_bytes(max_rows)
        dev = torch.device("cuda", torch.cuda.current_device())
        blobs = torch.zeros((len(images), blk), dtype=torch.uint8, device=dev)
        rows = torch.tensor([self.rows_of(im) for im in images], dtype=torch.int32, device=dev)
        keep = {}
        for k, (desc, kp) in enumerate(images):
            d = torch.as_tensor(np.ascontiguousarray(desc) if isinstance(desc, np.ndarray) else desc).to(dev)
            p = torch.as_tensor(np.ascontiguousarray(kp, dtype=np.float32) if isinstance(kp, np.ndarray)
                                else kp).to(dev).contiguous()
            d = d.contiguous()
            if d.dtype not in (torch.uint8, torch.float32):
                raise TypeError("descriptors must be float32 or uint8")
            if d.shape[0] > 0:
                keep.setdefault(d.dtype, []).append((k, d, p))
        self._torch_to_library()
        for dt, items in keep.items():  # one launch per element type
            matcher.prepare_many(self.ctx, [d.data_ptr() for _, d, _ in items], [d.shape[0] for _, d, _ in items],
                                 128 * items[0][1].element_size(),
                                 matcher.DESC_U8 if dt == torch.uint8 else matcher.DESC_F32,
                                 [p.data_ptr() for _, _, p in items], [blobs[k].data_ptr() for k, _, _ in items])
        matcher.check_inputs(self.ctx)  # waits for the conversion, so `keep` may go and torch may read
        return blobs, rows

    def match(self, blobs, rows, pairs):
        from . import matcher
        pairs = np.asarray(pairs).reshape(-1, 2)
        used = np.unique(pairs)
        self._torch_to_library()  # the gathered blocks were written on torch's (or RCCL's) stream
        base, stride = blobs.data_ptr(), blobs.stride(0) * blobs.element_size()
        imgs = [matcher.WrappedImage(self.ctx, base + int(g) * stride, int(rows[g])) for g in used]
        local_pairs = np.searchsorted(used, pairs).astype(np.int32)
        try:
            ng, ni = matcher.match_pairs(self.ctx, imgs, local_pairs, self.match_kwargs.get("K"),
                                         **{k: v for k, v in self.match_kwargs.items() if k != "K"})
        finally:
            for im in imgs:
                im.close()
        return ng, ni
