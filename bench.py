#!/usr/bin/env python3
"""Benchmark of the matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload: image pairs of 8192 x 8192 synthetic SIFT descriptors; per pair
kNN(k=2) -> ratio test (0.7) -> fundamental-matrix
RANSAC with 2048 hypotheses (fixed count) -> inlier selection.  One step
processes a batch of independent pairs (--pairs-per-step, default 64) whose
float32 descriptors and keypoints are already resident in HBM; the conversion
to the int8 layout is inside the timed region.  With N > 1 every rank runs its
own batch (pairs are independent units: weak scaling, no data-path collective);
ranks synchronise with a barrier on both sides of the timed region and the
slowest rank's time is used.

Prints one JSON line (rank 0).  `roofline` prices the distance kernel
(knn2_mfma) against the dense int8 MFMA peak; `cpu_baseline` is the CPU oracle
timed on this host's cores on one full pair of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_DENSE_PEAK_TOPS = 5000.0  # MI355X dense int8 MFMA: 2x the bf16 rate (~2.5 PF)


def make_pairs(n_pairs, n_desc, seed0, wrong_frac=0.0):
    from sfm_amd import synth
    descs, kps = [], []
    for p in range(n_pairs):
        s = seed0 + 10 * p
        Q, T, planted = synth.planted_pair(n_desc, n_desc, s + 1, s + 2, s + 3)
        kq, kt = synth.keypoints_for_pair(n_desc, n_desc, planted, s + 4, wrong_frac=wrong_frac)
        descs += [Q, T]
        kps += [kq, kt]
    return descs, kps


def cpu_baseline(descs, kps, n_hyp, threads, n_pairs):
    """CPU oracle on the first n_pairs pairs of the batch: kNN + ratio + F-RANSAC."""
    from oracle import pyoracle as po
    t_knn = t_ransac = 0.0
    n_good = n_inl = 0
    for p in range(n_pairs):
        t0 = time.perf_counter()
        m, c = po.knn_match(descs[2 * p], descs[2 * p + 1], 2, nthreads=threads)
        keep = po.ratio_test(m, c, 0.7)
        good = m[keep, 0]
        p1 = kps[2 * p][good["queryIdx"]].astype(np.float64)
        p2 = kps[2 * p + 1][good["trainIdx"]].astype(np.float64)
        t1 = time.perf_counter()
        r = po.ransac(p1, p2, None, model=po.MODEL_FUNDAMENTAL, prob=0.99, threshold=3.0,
                      fixed_count=n_hyp)
        t2 = time.perf_counter()
        t_knn += t1 - t0
        t_ransac += t2 - t1
        n_good += len(good)
        n_inl += int(r["n_inliers"])
    return dict(seconds=t_knn + t_ransac, knn_seconds=t_knn, ransac_seconds=t_ransac, n_pairs=n_pairs,
                n_good=n_good / n_pairs, n_inliers=n_inl / n_pairs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs-per-step", type=int, default=64)
    ap.add_argument("--n-desc", type=int, default=8192)
    ap.add_argument("--hyps", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wrong-frac", type=float, default=0.07,
                    help="share of descriptor matches that are geometrically wrong (reference log: 4-11 %%)")
    ap.add_argument("--no-microkernel", action="store_true", help="skip the MFMA-only ceiling launches")
    ap.add_argument("--knn-variant", type=int, default=-1, help="debug: force the sweep kernel shape")
    ap.add_argument("--score-filter", type=int, default=1, help="debug: 0 = double precision only")
    ap.add_argument("--score-prune", type=int, default=1, help="debug: 0 = score every model on every point")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2],
                    help="steps kept on the stream at once (1: collect each step before enqueueing the next)")
    ap.add_argument("--prep-stream", type=int, default=1, choices=[0, 1],
                    help="1: descriptor conversion of the next step on a side stream, beside the current step's "
                         "RANSAC (never beside a sweep); 0: everything on one stream")
    ap.add_argument("--no-overlap-probe", action="store_true",
                    help="skip the two-stream reference measurement after the timed region")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="contexts / HIP streams that consecutive steps alternate over (1: one stream, "
                         "kernels never overlap and their durations are their own)")
    ap.add_argument("--desc-u8", action="store_true",
                    help="descriptors arrive as CV_8U instead of cv::SIFT's CV_32F (not the headline configuration)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sfm_amd import matcher

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    # SFM_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with
    # fewer GPUs than ranks (ranks then share devices); the real run is RCCL.
    backend = os.environ.get("SFM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    B, n, nh = args.pairs_per_step, args.n_desc, args.hyps
    descs, kps = make_pairs(B, n, 1000 + 100000 * rank, args.wrong_frac)
    row_bytes = 128 * (1 if args.desc_u8 else 4)  # densely packed rows
    desc_dtype = matcher.DESC_U8 if args.desc_u8 else matcher.DESC_F32
    pairs = np.array([[2 * p, 2 * p + 1] for p in range(B)], np.int32)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        host_desc = np.stack(descs)
        if args.desc_u8:
            host_desc = host_desc.astype(np.uint8)
        d_desc = torch.from_numpy(host_desc).cuda()                # [2B, n, 128] float32 (or uint8)
        d_kp = torch.from_numpy(np.stack(kps)).cuda()              # [2B, n, 2] float32
    stream.synchronize()
    desc_ptrs = [d_desc[i].data_ptr() for i in range(2 * B)]
    kp_ptrs = [d_kp[i].data_ptr() for i in range(2 * B)]
    blk = matcher.image_packed_bytes(n)

    class Lane:
        """One matching context on its own stream.  A step (conversion + matching
        of one batch) belongs to one lane.  By default the conversion of step
        k+1 is enqueued on a side stream into a second set of packed images and
        ordered behind the front of step k (sweep and neighbour extraction), so
        that it runs beside the ratio test and the RANSAC of step k (HBM-bound
        next to latency- and VALU-bound work) and never beside a sweep:
        the sweep still has the chip to itself and its event-timed duration is
        its own.  --streams 2 alternates whole steps
        over two such lanes with no ordering between them (more pairs/s, but
        kernels of different steps then share the chip and their individual
        durations stop meaning anything)."""

        def __init__(self, st):
            self.stream = st
            self.n_sets = 2 if args.prep_stream else 1
            self.side = torch.cuda.Stream() if args.prep_stream else st
            self.ctx = matcher.Context(local_rank, stream=st.cuda_stream)
            self.ctx_prep = matcher.Context(local_rank, stream=self.side.cuda_stream) if args.prep_stream else self.ctx
            if args.knn_variant >= 0:
                self.ctx._L.sfm_gpu_debug_set_knn_variant(self.ctx.handle, args.knn_variant)
            self.packed, self.images, self.prep, self.batch, self.ready = [], [], [], [], []
            for _ in range(self.n_sets):
                with torch.cuda.stream(st):
                    d_packed = torch.empty((2 * B, blk), dtype=torch.uint8, device="cuda")
                st.synchronize()
                packed_ptrs = [d_packed[i].data_ptr() for i in range(2 * B)]
                images = [matcher.WrappedImage(self.ctx, packed_ptrs[i], n) for i in range(2 * B)]
                self.packed.append(d_packed)
                self.images.append(images)
                # argument arrays of the calls are built once (a C++ host holds them anyway)
                self.prep.append(matcher.PrepareBatch(self.ctx_prep, desc_ptrs, [n] * (2 * B), row_bytes, desc_dtype,
                                                      kp_ptrs, packed_ptrs))
                self.batch.append(matcher.PairBatch(self.ctx, images, pairs, None, ratio=0.7,
                                                    model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED,
                                                    prob=0.99, threshold=3.0, n_hyp=nh))
                self.ready.append(torch.cuda.Event())
            self.k = 0
            self.pending = []

        def begin_step(self):
            s = self.k % self.n_sets
            self.k += 1
            if args.prep_stream:
                matcher.stream_wait_front(self.ctx_prep, self.ctx)   # beside the previous step's RANSAC, not its sweep
                self.prep[s].run()
                self.ready[s].record(self.side)
                self.stream.wait_event(self.ready[s])
            else:
                self.prep[s].run()
            self.batch[s].begin()
            self.pending.append(s)

        def end_step(self):
            return self.batch[self.pending.pop(0)].end()

    lanes = [Lane(stream if i == 0 else torch.cuda.Stream()) for i in range(args.streams)]
    ctx, images, prep = lanes[0].ctx, lanes[0].images[0], lanes[0].prep[0]
    for lane in lanes:
        if args.score_filter != 1:
            lane.ctx.check(lane.ctx._L.sfm_gpu_debug_set_score_filter(lane.ctx.handle, args.score_filter))
        if args.score_prune != 1:
            lane.ctx.check(lane.ctx._L.sfm_gpu_debug_set_score_prune(lane.ctx.handle, args.score_prune))

    def step():
        prep.run()
        lanes[0].ctx_prep.synchronize()
        return matcher.match_pairs(ctx, images, pairs, None, ratio=0.7,
                                   model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED,
                                   prob=0.99, threshold=3.0, n_hyp=nh)

    # The timed loop keeps up to `--in-flight` steps on a lane's stream: step k+1
    # (conversion + matching) is enqueued before the host collects the counts of
    # step k, so the GPU does not idle through the host's round trip.  Every
    # step still runs start to finish, in order, on one stream.
    def run_steps(k, lanes=lanes):
        res, pending = None, []
        for i in range(k):
            lane = lanes[i % len(lanes)]
            lane.begin_step()
            pending.append(lane)
            if len(pending) == args.in_flight * len(lanes):
                res = pending.pop(0).end_step()
        while pending:
            res = pending.pop(0).end_step()
        return res

    if args.warmup:
        n_good, n_inl = run_steps(args.warmup)
    for lane in lanes:
        matcher.check_inputs(lane.ctx_prep)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: HIP events only around the dominant kernel (roofline figure)
    for lane in lanes:
        lane.ctx.profile_select("knn2_mfma")
        lane.ctx.profile_enable(True)
        lane.ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    n_good, n_inl = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    knn_ms, knn_cnt = 0.0, 0
    for lane in lanes:
        ms, cnt = lane.ctx.profile_query("knn2_mfma")
        knn_ms += ms
        knn_cnt += cnt
        if lane.ctx is not ctx:
            lane.ctx.profile_enable(False)
    # per-kernel breakdown from two more steps outside the timed region
    ctx.profile_select(None)
    ctx.profile_reset()
    pctx = lanes[0].ctx_prep
    if pctx is not ctx:
        pctx.profile_select(None)
        pctx.profile_enable(True)
        pctx.profile_reset()
    for _ in range(2):
        step()
    ctx.synchronize()
    kern = {}
    for name in ("prep_rows", "knn2_mfma", "knn_finalize", "knn_redo", "ratio_compact", "ransac_prep",
                 "ransac_table", "ransac_solve", "ransac_score", "ransac_select", "ransac_mask",
                 "inlier_compact"):
        ms, cnt = (pctx if name == "prep_rows" else ctx).profile_query(name)
        if cnt:
            kern[name] = {"avg_us": round(ms / cnt * 1e3, 2), "launches": cnt}
    if pctx is not ctx:
        pctx.profile_enable(False)
    # For reference, outside the timed region: the same steps alternating over
    # two unordered streams.  More pairs/s, but the sweep then shares the chip with
    # the other step's kernels and its own duration grows: not the headline.
    probe = None
    if args.streams == 1 and not args.no_overlap_probe:
        two = [lanes[0], Lane(torch.cuda.Stream())]
        two[1].ctx.profile_select("knn2_mfma")
        two[1].ctx.profile_enable(True)
        ctx.profile_select("knn2_mfma")
        ctx.profile_enable(True)
        run_steps(6, two)
        torch.cuda.synchronize()
        ctx.profile_reset()
        two[1].ctx.profile_reset()
        k2 = max(10, min(args.steps, 40))
        t1 = time.perf_counter()
        run_steps(k2, two)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        ms_a, cnt_a = ctx.profile_query("knn2_mfma")
        ms_b, cnt_b = two[1].ctx.profile_query("knn2_mfma")
        ctx.profile_enable(False)
        two[1].ctx.profile_enable(False)
        probe = {"pairs_per_s": round(B * k2 / dt, 1), "ms_per_step": round(dt / k2 * 1e3, 3), "steps": k2,
                 "knn2_avg_launch_us": round((ms_a + ms_b) / max(cnt_a + cnt_b, 1) * 1e3, 2),
                 "what": "this rank, steps alternating over two unordered streams (bench.py --streams 2); "
                         "kernels of different steps overlap, so per-kernel durations are not the kernels' own"}

    # ceiling on this box: the same sweep with its MFMAs and LDS fragment reads
    # only (no digest, no DMA, no barrier; results are garbage and unused).  That
    # kernel exists in the timing build of the library only, loaded here, after
    # the timed region, for this one measurement.
    ceil_ms, ceil_cnt = 0.0, 0
    if not args.no_microkernel:
        tctx = matcher.Context(local_rank, stream=stream.cuda_stream, timing_build=True)
        timages = [matcher.WrappedImage(tctx, lanes[0].packed[0][i].data_ptr(), n) for i in range(2 * B)]
        tctx.profile_select("knn2_mfma")
        tctx.profile_enable(True)
        tctx.check(tctx._L.sfm_gpu_debug_set_knn_variant(
            tctx.handle, (args.knn_variant if args.knn_variant >= 0 else 4) + 16 * 7))
        for _ in range(3):
            matcher.match_pairs(tctx, timages, pairs, None, ratio=0.7, model=matcher.MODEL_FUNDAMENTAL,
                                mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=nh)
        tctx.synchronize()
        ceil_ms, ceil_cnt = tctx.profile_query("knn2_mfma")
        for im in timages:
            im.close()
        tctx.close()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_pairs = B * args.steps * world
        value = total_pairs / elapsed
        knn = {"avg_us": round(knn_ms / max(knn_cnt, 1) * 1e3, 2), "launches": knn_cnt}
        ops_per_launch = 2.0 * n * n * 128 * B
        achieved = ops_per_launch / (knn["avg_us"] * 1e-6) / 1e12
        traffic = None
        pipe_busy = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_knn2_mfma.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("pairs_per_launch") == B and pmc.get("n_desc") == n:
                    traffic = pmc.get("hbm_bytes_per_launch")
                    pipe_busy = pmc.get("mfma_pipe_busy_frac")
            except Exception:
                traffic = None
        out = {
            "metric": "matched image pairs/sec (8k x 8k SIFT, k=2+RANSAC)",
            "value": round(value, 2),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int8 MFMA / int32 accumulate (kNN), f64 (RANSAC)",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: pairs of %dx%d SIFT-like descriptors (%s), kNN k=2 + ratio 0.7 + "
                            "fundamental RANSAC %d hyps (fixed count)" %
                            (n, n, "CV_8U" if args.desc_u8 else "CV_32F as cv::SIFT emits them", nh),
                "pairs_per_step": B * world,
                "pairs_per_gpu_per_step": B,
                "steps_in_flight": args.in_flight * args.streams,
                "streams": args.streams,
                "conversion_on_side_stream": bool(args.prep_stream),
                "n_good_mean": float(np.mean(n_good)),
                "n_inliers_mean": float(np.mean(n_inl)),
            },
            "roofline": {
                "kernel": "knn2_mfma",
                "bound": "mfma",
                "achieved": round(achieved, 1),
                "peak": INT8_DENSE_PEAK_TOPS,
                "unit": "TFLOP/s",
                "unit_note": "int8 multiply-add counted as 2 ops (TOP/s); peak = dense int8 MFMA",
                "frac": round(achieved / INT8_DENSE_PEAK_TOPS, 4),
                "traffic": traffic,
                "mfma_pipe_busy": {
                    "frac": pipe_busy,
                    "what": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) from profiles/pmc_knn2_mfma.json: "
                            "the share of the kernel's own clock cycles in which the matrix pipe is busy; frac above "
                            "is against the nominal 2.4 GHz peak, the clock under this load is lower",
                },
                "ops_per_launch": ops_per_launch,
                "avg_launch_us": knn["avg_us"],
                "mfma_only_microkernel": {
                    "what": "same launch with MFMAs + LDS fragment reads only, measured after the timed region",
                    "avg_launch_us": round(ceil_ms / max(ceil_cnt, 1) * 1e3, 2),
                    "frac_of_peak": round(ops_per_launch / (ceil_ms / max(ceil_cnt, 1) * 1e-3) / 1e12
                                          / INT8_DENSE_PEAK_TOPS, 4) if ceil_cnt else None,
                },
            },
            "kernels_outside_timed_region": kern,
            "two_unordered_streams_outside_timed_region": probe,
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg belongs to the single-GPU line only
            threads = int(os.environ.get("OMP_NUM_THREADS", "8"))
            threads = max(1, min(threads, 64))
            cb = cpu_baseline(descs, kps, nh, threads, min(B, 16))
            out["cpu_baseline"] = {
                "value": round(cb["n_pairs"] / cb["seconds"], 4),
                "unit": "pairs/s",
                "cores": threads,
                "kind": "port",
                "sample": "%d full pairs of the batch (%dx%d kNN on %d threads + ratio + F-RANSAC %d hyps on 1 "
                          "thread): %.2f s kNN, %.2f s RANSAC; inliers/pair %.1f (GPU: %.1f)" %
                          (cb["n_pairs"], n, n, threads, nh, cb["knn_seconds"], cb["ransac_seconds"],
                           cb["n_inliers"], float(np.mean(n_inl[:cb["n_pairs"]]))),
            }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for im in images:
        im.close()
    ctx.close()


if __name__ == "__main__":
    main()
