"""Exhaustive matching of V synthetic views (the shape of the exhaustive benchmark configuration:
64 views x 20k descriptors = 2016 pairs) through sfm_amd.allpairs: prepare where
the images live, all-gather the packed blocks (when launched under torchrun with
several ranks), deal the pairs, match, all-reduce the count matrices.

    python scripts/allpairs_perf.py [V] [n_desc] [E|F]
    python -m torch.distributed.run --nproc-per-node N ... scripts/allpairs_perf.py ...
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    model = sys.argv[3] if len(sys.argv) > 3 else "E"
    import torch
    import torch.distributed as dist
    from sfm_amd import allpairs, matcher, synth
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    torch.cuda.init()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    descs, kps, _ = synth.image_sequence(V, n, 4242, shared=0.4)
    lo, hi = allpairs.owned_range(V, world, rank)
    local = [(descs[i], kps[i]) for i in range(lo, hi)]
    ctx = matcher.Context(local_rank)
    if model == "E":
        be = allpairs.GpuBackend(ctx, K=synth.REF_K, model=matcher.MODEL_ESSENTIAL, prob=0.99, threshold=1.0)
    else:
        be = allpairs.GpuBackend(ctx, K=None, model=matcher.MODEL_FUNDAMENTAL, prob=0.99, threshold=3.0)
    dev = torch.device("cuda", local_rank)
    for rep in range(2):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        pairs, good, inl = allpairs.match_all_pairs(local, V, be, mode="all", device=dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            adj = [inl[i, i + 1] for i in range(V - 1)]
            print("run %d: %d views x %d descriptors, %d pairs on %d GPU(s): %.1f ms = %.0f pairs/s "
                  "(upload + conversion + exchange + match + RANSAC %s adaptive); inliers of neighbouring views "
                  "%d..%d, of the farthest pair %d" %
                  (rep, V, n, len(pairs), world, dt * 1e3, len(pairs) / dt, model, min(adj), max(adj), inl[0, V - 1]),
                  flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


main()
