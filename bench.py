#!/usr/bin/env python3
"""Benchmark of the matching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload: image pairs of 8192 x 8192 synthetic SIFT descriptors; per pair
kNN(k=2) -> ratio test (0.7) -> fundamental-matrix
RANSAC with 2048 hypotheses (fixed count) -> inlier selection.  One step
processes a batch of independent pairs (--pairs-per-step, default 64) whose
float32 descriptors and keypoints are already resident in HBM; the conversion
to the int8 layout is inside the timed region, and so is the path's product:
every pair's inlier DMatch records (the selectMatches of main.cpp:225-231) are
left in device memory (`with_records_to_host` in the line is the same loop with
those records also copied into pinned host memory behind each batch).
Consecutive steps work on different batches (--wrong-fracs: three input sets
whose share of geometrically wrong matches differs), so nothing the library
remembers from an earlier step describes the step it is sizing.
With N > 1 every rank runs its
own batch (pairs are independent units: weak scaling, no data-path collective);
ranks synchronise with a barrier on both sides of the timed region and the
slowest rank's time is used.

Ranks: one process per GPU.  Under a launcher (WORLD_SIZE set, e.g.
torch.distributed.run) this process is one rank.  Started plainly with
--gpus N > 1 it starts N fresh child processes itself, before anything touches
a GPU, one device each, and passes rank 0's line through; `n_gpus` in the line
is the number of ranks that joined the process group.

Prints one JSON line (rank 0).  `roofline` prices the distance kernel
(knn2_mfma) against the dense int8 MFMA peak; `roofline_secondary` the RANSAC
scoring pass; `configs2` one 32768 x 32768 pair with 4096 five-point
hypotheses; `allpairs` the exhaustive matching of 64 views x 20 000 descriptors
sharded over the ranks (strong scaling: all-gather of the prepared images,
dealt pair list, all-reduce of the count matrices); `cpu_baseline` is the CPU
oracle timed on this host's cores on full pairs of the same workload.
"""
import argparse
import contextlib
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_DENSE_PEAK_TOPS = 5000.0  # MI355X dense int8 MFMA: 2x the bf16 rate (~2.5 PF)


@contextlib.contextmanager
def no_gc_pause():
    """A timed region without the interpreter's cycle collector: a full collection of this process takes
    30-40 ms (seen as one such _end call among fifty of 0.7 ms), which is none of the measured work."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


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
        # byte rows + int32 sums: same records as the float32 restatement (tests/test_oracle_kat.py)
        m, c = po.knn_match(descs[2 * p], descs[2 * p + 1], 2, nthreads=threads, u8path=True)
        keep = po.ratio_test(m, c, 0.7)
        good = m[keep, 0]
        p1 = kps[2 * p][good["queryIdx"]].astype(np.float64)
        p2 = kps[2 * p + 1][good["trainIdx"]].astype(np.float64)
        t1 = time.perf_counter()
        r = po.ransac_fixed_mt(p1, p2, None, model=po.MODEL_FUNDAMENTAL, threshold=3.0, fixed_count=n_hyp,
                               nthreads=threads)
        t2 = time.perf_counter()
        t_knn += t1 - t0
        t_ransac += t2 - t1
        n_good += len(good)
        n_inl += int(r["n_inliers"])
    return dict(seconds=t_knn + t_ransac, knn_seconds=t_knn, ransac_seconds=t_ransac, n_pairs=n_pairs,
                n_good=n_good / n_pairs, n_inliers=n_inl / n_pairs)


def host_threads():
    """Threads for the CPU leg: the cores this process may run on, never more than
    the thread budget the environment sets (OMP_NUM_THREADS on a shared box)."""
    n = len(os.sched_getaffinity(0))
    try:
        n = min(n, int(os.environ["OMP_NUM_THREADS"]))
    except (KeyError, ValueError):
        pass
    return max(1, min(n, 64))


def spawn_ranks(n):
    """Starts n fresh rank processes of this script (nothing in this process has
    touched a GPU) and returns their exit code: 0 when all end with 0; when one
    fails, the others are terminated at once (they would otherwise wait for it in
    a rendezvous or a collective until some outer time limit) and its code is
    returned.  Rank 0 inherits stdout, so its JSON line is this command's output."""
    import torch
    backend = os.environ.get("SFM_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not os.environ.get("SFM_BENCH_DRYRUN"):
        have = torch.cuda.device_count()  # counting devices does not initialise the runtime
        if have < n:
            raise SystemExit("--gpus %d needs %d GPUs, %d visible (SFM_BENCH_BACKEND=gloo rehearses the "
                             "control flow with ranks sharing devices)" % (n, n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        # The hosts this runs on share GPU memory between processes through dmabuf
        # only; with the legacy IPC mode RCCL's set-up fails in hipIpcGetMemHandle
        # ("invalid argument").  The machine images export this already: keep it
        # for the ranks whatever the caller's environment was.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = abs(code) or 1
                print("bench.py: rank %d exited with %d; stopping the other ranks" % (procs.index(p), code),
                      file=sys.stderr, flush=True)
                break
    for p in live:   # only non-empty after a failure
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


def source_sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


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
                    help="share of descriptor matches that are geometrically wrong in the one-pair block "
                         "(reference log: 4-11 %%)")
    ap.add_argument("--wrong-fracs", type=str, default="0.07,0.25,0.12",
                    help="one input set per value (different pairs, this share of geometrically wrong matches); "
                         "consecutive steps rotate over the sets, so with two steps in flight no step is sized "
                         "from a step on its own data")
    ap.add_argument("--records", type=int, default=1, choices=[0, 1],
                    help="1: every step leaves each pair's inlier DMatch records in device memory (what match() "
                         "returns); 0: counts only (the match-count matrix use)")
    ap.add_argument("--deadline", type=int, default=1500,
                    help="seconds after which a rank that is still running gives up with exit code 3 (a rendezvous "
                         "or collective that never completes must not outlive the caller's patience)")
    ap.add_argument("--no-microkernel", action="store_true", help="skip the MFMA-only ceiling launches")
    ap.add_argument("--knn-variant", type=int, default=-1, help="debug: force the sweep kernel shape")
    ap.add_argument("--score-filter", type=int, default=1, help="debug: 0 = double precision only")
    ap.add_argument("--score-prune", type=int, default=1, help="debug: 0 = score every model on every point")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2],
                    help="steps kept on the stream at once (1: collect each step before enqueueing the next)")
    ap.add_argument("--prep-stream", type=int, default=1, choices=[0, 1],
                    help="1: descriptor conversion of the next step on a side stream, beside the current step's "
                         "RANSAC (never beside a sweep); 0: everything on one stream")
    ap.add_argument("--main-priority", type=int, default=0,
                    help="HIP priority of the stream the matching runs on (-1: above the conversion's side stream)")
    ap.add_argument("--no-overlap-probe", action="store_true",
                    help="skip the two-stream reference measurement after the timed region")
    ap.add_argument("--no-records-to-host", action="store_true",
                    help="skip the block that repeats the loop with the records also copied to the host (a kernel "
                         "trace of the run then holds the timed loop's launches only: under the profiler that copy is "
                         "a shader kernel beside the sweep)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="contexts / HIP streams that consecutive steps alternate over (1: one stream, "
                         "kernels never overlap and their durations are their own)")
    ap.add_argument("--desc-u8", action="store_true",
                    help="descriptors arrive as CV_8U instead of cv::SIFT's CV_32F (not the headline configuration)")
    ap.add_argument("--no-configs2", action="store_true", help="skip the 32768 x 32768 pair block")
    ap.add_argument("--no-allpairs", action="store_true", help="skip the exhaustive-matching block")
    ap.add_argument("--extras-deadline", type=int, default=420,
                    help="seconds the blocks after the timed region may take before the line is printed without them")
    ap.add_argument("--allpairs-views", type=int, default=64)
    ap.add_argument("--allpairs-desc", type=int, default=20000)
    ap.add_argument("--allpairs-window", type=int, default=0,
                    help="0: every view sees one common scene (the 64-view configuration); w > 0: a long capture in "
                         "which view i overlaps views i-w..i+w only (the 1000-image configuration: --allpairs-views "
                         "1000 --allpairs-desc 8192 --allpairs-window 5)")
    ap.add_argument("--allpairs-reps", type=int, default=3, help="runs of the exhaustive block (the first sizes workspaces)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # one process per GPU: start the ranks before anything here touches a GPU
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    # One line per run, whoever prints it (the main thread or a watchdog).
    out = {}
    out_lock = threading.Lock()
    printed = [False]
    stage = ["start"]

    def print_line_once(extra=None):
        with out_lock:
            if printed[0] or rank != 0 or not out:
                return
            printed[0] = True
            line = dict(out)
            if extra:
                line.update(extra)
            print(json.dumps(line), flush=True)

    # Running from before the rendezvous to the end: a rank whose peers never
    # arrive, or a collective that never completes, ends here with a non-zero
    # code (rank 0 first prints whatever part of the line it has).
    def out_of_time():
        print("bench.py rank %d: deadline of %d s passed in: %s" % (rank, args.deadline, stage[0]),
              file=sys.stderr, flush=True)
        print_line_once({"incomplete": "deadline of %d s passed in: %s" % (args.deadline, stage[0])})
        os._exit(3)

    deadline = threading.Timer(args.deadline, out_of_time)
    deadline.daemon = True
    deadline.start()

    import datetime

    import torch
    import torch.distributed as dist

    pg_timeout = datetime.timedelta(seconds=max(60, min(args.deadline, 600)))
    if os.environ.get("SFM_BENCH_DRYRUN"):
        # control flow only (no GPU work): rendezvous, barrier, reduction, one line
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            stage[0] = "rendezvous"
            if os.environ.get("SFM_BENCH_TEST_EXIT_RANK") == str(rank):
                sys.exit(7)   # rehearsal of a rank that dies before the rendezvous
            dist.init_process_group("gloo", timeout=pg_timeout)
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            joined = dist.get_world_size()
            assert t.item() == world
        else:
            joined = 1
        if rank == 0:
            print(json.dumps({"metric": "matched image pairs/sec (8k x 8k SIFT, k=2+RANSAC)", "value": 0.0,
                              "unit": "pairs/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
                              "dry_run": True}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        deadline.cancel()
        return
    from sfm_amd import matcher
    # SFM_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with
    # fewer GPUs than ranks (ranks then share devices); the real run is RCCL.
    backend = os.environ.get("SFM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        stage[0] = "rendezvous"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    # how many ranks the collective library itself reaches: a sum of ones
    saw_ranks = 1
    if world > 1:
        stage[0] = "first all-reduce"
        ones = torch.ones(1, dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        saw_ranks = int(ones.item())
    stage[0] = "set-up"

    B, n, nh = args.pairs_per_step, args.n_desc, args.hyps
    fracs = [float(x) for x in args.wrong_fracs.split(",") if x.strip()]
    row_bytes = 128 * (1 if args.desc_u8 else 4)  # densely packed rows
    desc_dtype = matcher.DESC_U8 if args.desc_u8 else matcher.DESC_F32
    pairs = np.array([[2 * p, 2 * p + 1] for p in range(B)], np.int32)
    stream = torch.cuda.Stream(priority=args.main_priority)
    # One input set per entry of --wrong-fracs: different pairs (other seeds) with
    # that share of geometrically wrong matches.  Set 0 is also the CPU leg's.
    inputs = []
    descs = kps = None
    for si, wf in enumerate(fracs):
        d_s, k_s = make_pairs(B, n, 1000 + 100000 * rank + 7000 * si, wf)
        if si == 0:
            descs, kps = d_s, k_s
        with torch.cuda.stream(stream):
            host_desc = np.stack(d_s)
            if args.desc_u8:
                host_desc = host_desc.astype(np.uint8)
            d_desc = torch.from_numpy(host_desc).cuda()                # [2B, n, 128] float32 (or uint8)
            d_kp = torch.from_numpy(np.stack(k_s)).cuda()              # [2B, n, 2] float32
        stream.synchronize()
        inputs.append(dict(desc=d_desc, kp=d_kp, wrong_frac=wf,
                           desc_ptrs=[d_desc[i].data_ptr() for i in range(2 * B)],
                           kp_ptrs=[d_kp[i].data_ptr() for i in range(2 * B)]))
        del host_desc, d_s, k_s
    n_in = len(inputs)
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
            self.rec_dev, self.rec_host, self.batch_host = [], [], []
            kw = dict(ratio=0.7, model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED, prob=0.99,
                      threshold=3.0, n_hyp=nh)
            for _ in range(2):   # two batches may be in flight: two sets of output buffers
                with torch.cuda.stream(st):
                    # each pair's selectMatches: B x n cv::DMatch records (16 bytes each)
                    self.rec_dev.append(torch.zeros((B, n, 4), dtype=torch.int32, device="cuda"))
                self.rec_host.append(torch.zeros((B, n, 4), dtype=torch.int32).pin_memory())
            for si in range(self.n_sets):
                with torch.cuda.stream(st):
                    d_packed = torch.empty((2 * B, blk), dtype=torch.uint8, device="cuda")
                st.synchronize()
                packed_ptrs = [d_packed[i].data_ptr() for i in range(2 * B)]
                images = [matcher.WrappedImage(self.ctx, packed_ptrs[i], n) for i in range(2 * B)]
                self.packed.append(d_packed)
                self.images.append(images)
                # argument arrays of the calls are built once (a C++ host holds them anyway):
                # one conversion per (input set, packed set), one batch per (packed set, output set)
                self.prep.append([matcher.PrepareBatch(self.ctx_prep, inp["desc_ptrs"], [n] * (2 * B), row_bytes,
                                                       desc_dtype, inp["kp_ptrs"], packed_ptrs) for inp in inputs])
                if args.records:
                    self.batch.append([matcher.PairBatch(self.ctx, images, pairs, None, records_dev=r.data_ptr(),
                                                         match_capacity=n, **kw) for r in self.rec_dev])
                else:
                    self.batch.append([matcher.PairBatch(self.ctx, images, pairs, None, **kw)] * 2)
                self.batch_host.append([matcher.PairBatch(self.ctx, images, pairs, None, records_dev=r.data_ptr(),
                                                          match_capacity=n, records_host=h.data_ptr(), **kw)
                                        for r, h in zip(self.rec_dev, self.rec_host)])
                self.ready.append(torch.cuda.Event())
            self.k = 0
            self.pending = []

        def begin_step(self, to_host=False):
            s = self.k % self.n_sets      # packed images this step converts into
            i = self.k % n_in             # input set of this step
            o = self.k % 2                # output buffers of this step
            self.k += 1
            if args.prep_stream:
                matcher.stream_wait_front(self.ctx_prep, self.ctx)   # beside the previous step's RANSAC, not its sweep
                self.prep[s][i].run()
                self.ready[s].record(self.side)
                self.stream.wait_event(self.ready[s])
            else:
                self.prep[s][i].run()
            b = (self.batch_host if to_host else self.batch)[s][o]
            b.begin()
            self.pending.append((b, i, o, self.k))

        def end_step(self):
            b, i, o, k = self.pending.pop(0)
            g, c = b.end()
            return g, c, i, o, k

    lanes = [Lane(stream if i == 0 else torch.cuda.Stream()) for i in range(args.streams)]
    ctx, images = lanes[0].ctx, lanes[0].images[0]
    for lane in lanes:
        if args.score_filter != 1:
            lane.ctx.check(lane.ctx._L.sfm_gpu_debug_set_score_filter(lane.ctx.handle, args.score_filter))
        if args.score_prune != 1:
            lane.ctx.check(lane.ctx._L.sfm_gpu_debug_set_score_prune(lane.ctx.handle, args.score_prune))

    def step(i):
        lanes[0].prep[0][i % n_in].run()
        lanes[0].ctx_prep.synchronize()
        return matcher.match_pairs(ctx, images, pairs, None, ratio=0.7,
                                   model=matcher.MODEL_FUNDAMENTAL, mode=matcher.RANSAC_FIXED,
                                   prob=0.99, threshold=3.0, n_hyp=nh)

    # The timed loop keeps up to `--in-flight` steps on a lane's stream: step k+1
    # (conversion + matching) is enqueued before the host collects the counts of
    # step k, so the GPU does not idle through the host's round trip.  Every
    # step still runs start to finish, in order, on one stream.
    last = {}   # input set -> (n_good, n_inliers, output set) of its most recent step

    def run_steps(k, lanes=lanes, to_host=False):
        pending = []

        def collect():
            g, c, i, o, k = pending.pop(0).end_step()
            last[i] = (g, c, o, k)

        for i in range(k):
            lane = lanes[i % len(lanes)]
            lane.begin_step(to_host)
            pending.append(lane)
            if len(pending) == args.in_flight * len(lanes):
                collect()
        while pending:
            collect()

    if args.warmup:
        run_steps(args.warmup)
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
    with no_gc_pause():
        fence()
        stage[0] = "timed region"
        t0 = time.perf_counter()
        run_steps(args.steps)
        fence()
        elapsed = time.perf_counter() - t0
    stage[0] = "after the timed region"
    per_set = {i: (float(np.mean(v[0])), float(np.mean(v[1]))) for i, v in sorted(last.items())}
    n_good = np.concatenate([last[i][0] for i in sorted(last)])
    n_inl = np.concatenate([last[i][1] for i in sorted(last)])
    n_inl_set0 = last[0][1] if 0 in last else n_inl
    # the records of the last step on every input set, as the timed loop left them:
    # per pair n_inliers records, query rows ascending as match() emits them
    records_ok = None
    if args.records and args.streams == 1:
        records_ok = True
        torch.cuda.synchronize()
        for i, (g, c, o, k) in last.items():
            if k <= lanes[0].k - 2:
                continue   # that step's output buffers have been reused since
            rec = lanes[0].rec_dev[o].cpu().numpy()
            for p_ in range(B):
                q = rec[p_, :c[p_], 0]
                records_ok = records_ok and bool((np.diff(q) > 0).all()) and bool((rec[p_, :c[p_], 2] == 0).all())
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
    for it in range(2):
        step(it)
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
    ctx.profile_enable(False)   # (the blocks below that want per-launch events switch them on themselves)
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
        with no_gc_pause():
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

    # The same loop with each batch's records also copied to pinned host memory
    # behind the batch (on a side stream, beside the next batch's sweep): what a
    # host that consumes selectMatches on the CPU, as the reference's later stages
    # do, would see.
    to_host = None
    if args.records and args.streams == 1 and not args.no_records_to_host:
        stage[0] = "records to host"
        run_steps(4, to_host=True)
        torch.cuda.synchronize()
        k3 = max(10, min(args.steps, 50))
        with no_gc_pause():
            t1 = time.perf_counter()
            run_steps(k3, to_host=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
        ok = True
        for i, (g, c, o, k) in last.items():
            if k <= lanes[0].k - 2:
                continue
            h = lanes[0].rec_host[o].numpy()
            d = lanes[0].rec_dev[o].cpu().numpy()
            for p_ in range(B):
                ok = ok and bool((h[p_, :c[p_]] == d[p_, :c[p_]]).all())
        to_host = {"pairs_per_s": round(B * k3 / dt, 1), "ms_per_step": round(dt / k3 * 1e3, 3), "steps": k3,
                   "record_bytes_per_step": int(16 * sum(int(v[1].sum()) for v in last.values()) / max(len(last), 1)),
                   "host_copy_equals_device_records": ok,
                   "what": "this rank, after the timed region: the timed loop with every pair's inlier records "
                           "also copied to pinned host memory behind its batch, on a side stream"}

    # ceiling on this box: the same sweep with its MFMAs and LDS fragment reads
    # only (no digest, no DMA, no barrier; results are garbage and unused).  That
    # kernel exists in the timing build of the library only, loaded here, after
    # the timed region, for this one measurement.
    ceil_ms, ceil_cnt = 0.0, 0
    if not args.no_microkernel:
        try:
            tctx = matcher.Context(local_rank, stream=stream.cuda_stream, timing_build=True)
            timages = [matcher.WrappedImage(tctx, lanes[0].packed[0][i].data_ptr(), n) for i in range(2 * B)]
            tctx.profile_select("knn2_mfma")
            tctx.profile_enable(True)
            tctx.check(tctx._L.sfm_gpu_debug_set_knn_variant(
                tctx.handle, (args.knn_variant if args.knn_variant >= 0 else 4) + 16 * 7))
            for it in range(5):
                if it == 2:   # the first calls load the timing library's code objects and size its workspaces
                    tctx.profile_reset()
                matcher.match_pairs(tctx, timages, pairs, None, ratio=0.7, model=matcher.MODEL_FUNDAMENTAL,
                                    mode=matcher.RANSAC_FIXED, prob=0.99, threshold=3.0, n_hyp=nh)
            tctx.synchronize()
            ceil_ms, ceil_cnt = tctx.profile_query("knn2_mfma")
            for im in timages:
                im.close()
            tctx.close()
        except (OSError, RuntimeError) as e:   # no timing library in this tree: the line goes without the ceiling
            print("mfma-only microkernel skipped: %s" % e, file=sys.stderr, flush=True)
            ceil_ms, ceil_cnt = 0.0, 0
    ctx.profile_select(None)
    ctx.profile_enable(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    joined = dist.get_world_size() if world > 1 else 1

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
                # counters are only quoted for the kernel source they were collected on
                same_kernel = pmc.get("kernel_source_sha256") == source_sha(
                    os.path.join(ROOT, "sfm_amd", "csrc", "knn_kernels.hip"))
                if same_kernel and pmc.get("pairs_per_launch") == B and pmc.get("n_desc") == n:
                    traffic = pmc.get("hbm_bytes_per_launch")
                    pipe_busy = pmc.get("mfma_pipe_busy_frac")
            except Exception:
                traffic = None
        out.update({
            "metric": "matched image pairs/sec (8k x 8k SIFT, k=2+RANSAC)",
            "value": round(value, 2),
            "unit": "pairs/s",
            "n_gpus": joined,
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
                "output_per_pair": "inlier DMatch records in device memory (selectMatches of main.cpp:225-231) + "
                                   "(n_good, n_inliers) on the host" if args.records else "(n_good, n_inliers) only",
                "records_checked": records_ok,
                "input_sets": [{"wrong_frac": inputs[i]["wrong_frac"], "n_good_mean": per_set[i][0],
                                "n_inliers_mean": per_set[i][1]} for i in sorted(per_set)],
                "input_rotation": "step k works on input set k mod %d with %d steps in flight: the sizes the library "
                                  "remembers when it plans a step come from a step on another set" %
                                  (n_in, args.in_flight * args.streams),
                "n_good_mean": float(np.mean(n_good)),
                "n_inliers_mean": float(np.mean(n_inl)),
            },
            "roofline": {
                "kernel": "knn2_mfma",
                "bound": "mfma",
                "achieved": round(achieved, 1),
                "peak": INT8_DENSE_PEAK_TOPS,
                "unit": "TOP/s",
                "unit_note": "int8 multiply-add counted as 2 ops; peak = dense int8 MFMA",
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
                    "avg_launch_us": round(ceil_ms / ceil_cnt * 1e3, 2) if ceil_cnt else None,
                    "frac_of_peak": round(ops_per_launch / (ceil_ms / max(ceil_cnt, 1) * 1e-3) / 1e12
                                          / INT8_DENSE_PEAK_TOPS, 4) if ceil_cnt else None,
                },
            },
            "with_records_to_host": to_host,
            "rccl_saw_ranks": saw_ranks if world > 1 and backend == "nccl" else None,
            "collective_saw_ranks": saw_ranks,
            "kernels_outside_timed_region": kern,
            "two_unordered_streams_outside_timed_region": probe,
        })
        sc = kern.get("ransac_score")
        if sc:
            # RANSAC scoring, second kernel of the step: every candidate model against every
            # ratio-test survivor.  Algorithmic work = the exhaustive loop's (models x points
            # tests of ~40 double-precision flops, models*72 B + points*32 B read once); the
            # kernel settles most tests in float32 and drops models that cannot win, so it
            # does less than that.  Points stay in L2 / the scalar cache: the bound is vector
            # issue, not HBM.
            n_models = 3.0 * nh * 0.83        # seven-point solver: up to 3 real roots, ~2.5 on average
            tests = B * n_models * float(np.mean(n_good))
            step_us = sc["avg_us"] * sc["launches"] / 2.0   # launches counted over the two extra steps
            bytes_alg = B * (n_models * 72.0 + float(np.mean(n_good)) * 32.0)
            out["roofline_secondary"] = {
                "kernel": "ransac_score", "bound": "vector issue (packed float32 filter, double precision for "
                                                   "the ~1 % of tests the filter cannot settle)",
                "us_per_step": round(step_us, 2),
                "algorithmic_tests_per_step": round(tests),
                "algorithmic_gtests_per_s": round(tests / (step_us * 1e-6) / 1e9, 1),
                "equivalent_fp64_tflops_at_40_flops_per_test": round(tests * 40.0 / (step_us * 1e-6) / 1e12, 1),
                "fp64_vector_peak_tflops": 78.6,
                "hbm": {"algorithmic_bytes_per_step": round(bytes_alg), "achieved_GB_s": round(bytes_alg / (step_us * 1e-6) / 1e9, 1),
                        "peak_GB_s": 8000.0, "frac": round(bytes_alg / (step_us * 1e-6) / 8e12, 4)},
                "note": "the equivalent rate exceeds the FP64 peak because the exact staged bound scores most "
                        "models on a fraction of the points; counters for this kernel: profiles/",
            }
    # The further blocks (outside the timed region) run under a deadline of their
    # own: if one of them hangs on any rank -- a collective that never completes, a
    # device that stopped answering -- every rank leaves when it passes; rank 0 first
    # prints the line it has, so the headline measurement is not lost, and the exit
    # code (3) says that the run did not finish.
    def give_up():
        print("bench.py rank %d: the blocks after the timed region did not finish in %d s (in: %s)" %
              (rank, args.extras_deadline, stage[0]), file=sys.stderr, flush=True)
        print_line_once({"extras_incomplete": "deadline of %d s passed in: %s" % (args.extras_deadline, stage[0])})
        os._exit(3)

    watchdog = threading.Timer(args.extras_deadline, give_up)
    watchdog.daemon = True
    watchdog.start()
    configs2 = allp = None
    try:
        # ---- the one-large-pair configuration: one 32768 x 32768 pair + essential RANSAC, 4096 hypotheses
        # (outside the timed region; rank 0 of a single-GPU run only)
        stage[0] = "configs2"
        if rank == 0 and world == 1 and not args.no_configs2:
            from sfm_amd import synth
            n2, nh2 = 32768, 4096
            Q2, T2, planted2 = synth.planted_pair(n2, n2, 1101, 1102, 1103)
            kq2, kt2 = synth.keypoints_for_pair(n2, n2, planted2, 2101, wrong_frac=args.wrong_frac)
            with torch.cuda.stream(stream):
                d2 = torch.from_numpy(np.stack([Q2, T2])).cuda()
                k2 = torch.from_numpy(np.stack([kq2, kt2])).cuda()
                blk2 = matcher.image_packed_bytes(n2)
                pk2 = torch.empty((2, blk2), dtype=torch.uint8, device="cuda")
            stream.synchronize()
            matcher.prepare_many(ctx, [d2[i].data_ptr() for i in range(2)], [n2, n2], 512, matcher.DESC_F32,
                                 [k2[i].data_ptr() for i in range(2)], [pk2[i].data_ptr() for i in range(2)])
            matcher.check_inputs(ctx)
            im2 = [matcher.WrappedImage(ctx, pk2[i].data_ptr(), n2) for i in range(2)]
            one = np.array([[0, 1]], np.int32)
            kw2 = dict(ratio=0.7, model=matcher.MODEL_ESSENTIAL, mode=matcher.RANSAC_FIXED, prob=0.99, threshold=1.0,
                       n_hyp=nh2)
            from sfm_amd import synth as _synth
            for _ in range(3):
                g2, i2 = matcher.match_pairs(ctx, im2, one, _synth.REF_K, **kw2)
            # host clock per call first, without the per-launch events (creating and reading those is host
            # work of its own), then the same calls again with them for the kernels' durations
            reps2 = 10
            ctx.synchronize()
            with no_gc_pause():
                t2 = time.perf_counter()
                for _ in range(reps2):
                    g2, i2 = matcher.match_pairs(ctx, im2, one, _synth.REF_K, **kw2)
                ctx.synchronize()
                wall2 = (time.perf_counter() - t2) / reps2
            ctx.profile_select(None)
            ctx.profile_enable(True)
            ctx.profile_reset()
            for _ in range(reps2):
                g2, i2 = matcher.match_pairs(ctx, im2, one, _synth.REF_K, **kw2)
            ctx.synchronize()
            k2us = {}
            for name in ("knn2_mfma", "knn_finalize", "ransac_solve", "ransac_score", "ransac_select"):
                ms, cnt = ctx.profile_query(name)
                if cnt:
                    k2us[name] = round(ms / cnt * 1e3, 2)
            ctx.profile_enable(False)
            ops2 = 2.0 * n2 * n2 * 128
            configs2 = {
                "workload": "configs[2]: one pair of %dx%d descriptors, kNN k=2 + ratio 0.7 + five-point essential "
                            "RANSAC %d hyps (fixed count)" % (n2, n2, nh2),
                "knn2_mfma_us": k2us.get("knn2_mfma"),
                "knn2_mfma_frac_of_int8_peak": round(ops2 / (k2us["knn2_mfma"] * 1e-6) / 1e12 / INT8_DENSE_PEAK_TOPS, 4)
                if k2us.get("knn2_mfma") else None,
                "five_point_solve_us": k2us.get("ransac_solve"),
                "scoring_us": k2us.get("ransac_score"),
                "neighbour_extraction_us": k2us.get("knn_finalize"),
                "ms_per_pair_host_clock": round(wall2 * 1e3, 3),
                "n_good": int(g2[0]), "n_inliers": int(i2[0]),
                "what": "host clock: average of %d one-pair calls; kernel figures: HIP events around each launch of "
                        "%d further calls (one sweep launch of 512 workgroups); measured after the timed region"
                        % (reps2, reps2),
            }
            for im in im2:
                im.close()
            del d2, k2, pk2

        # ---- the exhaustive configuration: all pairs of V views, sharded over the ranks (strong scaling)
        stage[0] = "allpairs"
        if not args.no_allpairs:
            from sfm_amd import allpairs, synth
            V, nd = args.allpairs_views, args.allpairs_desc
            lo, hi = allpairs.owned_range(V, world, rank)
            if args.allpairs_window > 0:
                descs_v, kps_v = synth.window_sequence(lo, hi, nd, 4242, shared=0.4, window=args.allpairs_window)
                own = range(hi - lo)
            else:
                descs_v, kps_v, _ = synth.image_sequence(hi, nd, 4242, shared=0.4)   # images lo..hi-1 are this rank's
                own = range(lo, hi)
            with torch.cuda.stream(stream):
                local = [(torch.from_numpy(descs_v[i]).cuda(), torch.from_numpy(kps_v[i]).cuda()) for i in own]
            stream.synchronize()
            del descs_v, kps_v
            be = allpairs.GpuBackend(ctx, K=synth.REF_K, model=matcher.MODEL_ESSENTIAL, mode=matcher.RANSAC_ADAPTIVE,
                                     prob=0.99, threshold=1.0, max_iters=1000)
            dev = torch.device("cuda", local_rank)
            best = None
            with torch.cuda.stream(stream):
                for rep in range(max(2, args.allpairs_reps)):   # the first run sizes the workspaces
                    fence()
                    t3 = time.perf_counter()
                    ap_pairs, ap_good, ap_inl, ap_stats = allpairs.match_all_pairs(local, V, be, mode="all", device=dev,
                                                                                 return_stats=True)
                    fence()
                    dt3 = time.perf_counter() - t3
                    if world > 1:
                        tt = torch.tensor([dt3], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                        dt3 = float(tt.item())
                    if rep > 0:
                        best = dt3 if best is None else min(best, dt3)
            adj = [int(ap_inl[i, i + 1]) for i in range(V - 1)]
            per8 = (V + 7) // 8   # what one all-gather delivers to each of 8 ranks: their slabs of ceil(V/8) images
            allp = {
                "workload": "%s: %d views x %d descriptors, all %d pairs, kNN k=2 + ratio 0.7 + essential RANSAC "
                            "(adaptive, prob 0.99, 1 px), descriptors resident in HBM as float32" %
                            ("configs[4]" if args.allpairs_window > 0 else "configs[3]", V, nd, len(ap_pairs)),
                "sweep_tops_algorithmic": round(2.0 * nd * nd * 128 * len(ap_pairs) / best / 1e12, 1),
                "sweep_frac_of_int8_peak_end_to_end": round(2.0 * nd * nd * 128 * len(ap_pairs) / best / 1e12
                                                            / INT8_DENSE_PEAK_TOPS, 4),
                "gathered_bytes_per_rank_at_8_ranks": int(8 * per8 * (ap_stats["blob_bytes"] + 4)),
                "pairs_with_inliers": int((ap_inl > 0).sum()),
                "scaling": "strong", "n_gpus": joined, "pairs": int(len(ap_pairs)),
                "seconds": round(best, 5), "pairs_per_s": round(len(ap_pairs) / best, 1),
                "includes": "descriptor conversion where the images live, all-gather of the prepared images, dealt pair "
                            "list, matching, all-reduce of the V x V count matrices (max over ranks, best of %d runs)"
                            % (max(2, args.allpairs_reps) - 1),
                "gathered_bytes_per_rank": ap_stats["gathered_bytes"], "collective_backend": backend if world > 1 else None,
                "inliers_neighbouring_views_min_max": [min(adj), max(adj)] if adj else None,
                "inliers_farthest_pair": int(ap_inl[0, V - 1]) if V > 1 else None,
            }
            del local

    except Exception as e:  # noqa: BLE001 -- reported in the line and on stderr; the headline above stands
        print("bench.py rank %d: %s in %s: %s" % (rank, type(e).__name__, stage[0], e), file=sys.stderr, flush=True)
        if rank == 0:
            with out_lock:
                out["extras_error"] = "%s in %s: %s" % (type(e).__name__, stage[0], e)
    if rank == 0:
        with out_lock:
            if configs2:
                out["configs2"] = configs2
            if allp:
                out["allpairs"] = allp
    if world == 1:
        watchdog.cancel()
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # the CPU leg belongs to the single-GPU line only
            threads = host_threads()
            cb = cpu_baseline(descs, kps, nh, threads, min(B, 64))
            out["cpu_baseline"] = {
                "value": round(cb["n_pairs"] / cb["seconds"], 4),
                "unit": "pairs/s",
                "cores": threads,
                "kind": "port",
                "host_cpus_visible": os.cpu_count(),
                "sample": "%d full pairs of the batch (%dx%d kNN on bytes with int32 sums, threaded over query rows + ratio + F-RANSAC %d hyps "
                          "threaded over hypotheses, %d threads = this process's CPU share): %.2f s kNN, %.2f s "
                          "RANSAC; inliers/pair %.1f (GPU: %.1f)" %
                          (cb["n_pairs"], n, n, nh, threads, cb["knn_seconds"], cb["ransac_seconds"],
                           cb["n_inliers"], float(np.mean(n_inl_set0[:cb["n_pairs"]]))),
            }
    if world > 1:
        stage[0] = "final barrier"
        dist.barrier()
    watchdog.cancel()
    deadline.cancel()
    print_line_once()
    if world > 1:
        dist.destroy_process_group()
    for im in images:
        im.close()
    ctx.close()


if __name__ == "__main__":
    main()
