"""bench.py starts its own ranks: `--gpus N` launched plainly must run N
processes that join one process group, and the line must say how many did."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 8])
def test_plain_launch_spawns_the_ranks_control_flow_only(n):
    # no GPU work: rendezvous over gloo, barrier, reduction, rank 0 prints
    line = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"], {"SFM_BENCH_DRYRUN": "1"}, 240)
    assert line["n_gpus"] == n and line["dry_run"] is True


def test_a_rank_that_dies_before_the_rendezvous_ends_the_run_at_once():
    """One of three ranks exits before joining the process group: the command
    must stop the other two and fail within seconds, not leave them waiting in
    the rendezvous for a time limit somewhere above."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SFM_BENCH_DRYRUN="1", SFM_BENCH_TEST_EXIT_RANK="1")
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert "rank 1 exited with 7" in p.stderr
    assert time.time() - t0 < 60
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_a_rank_alone_in_the_rendezvous_gives_up_at_the_deadline():
    """A rank started by a launcher whose peers never come: the deadline ends it
    with exit code 3 instead of leaving it in the rendezvous."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SFM_BENCH_DRYRUN="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--deadline", "8"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 3 and "deadline of 8 s passed in: rendezvous" in p.stderr


def test_a_launcher_world_size_must_match_gpus():
    env = dict(os.environ, SFM_BENCH_DRYRUN="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_run_the_whole_bench(gpu_ctx):
    """Both ranks share device 0 and exchange over gloo (the real run is one GPU
    per rank over RCCL): the timed loop, the barriers and the sharded exhaustive
    matching with its all-gather and all-reduce all run for real."""
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs-per-step", "8", "--n-desc", "2048",
            "--hyps", "256", "--no-cpu-baseline", "--no-microkernel", "--no-overlap-probe", "--no-configs2",
            "--allpairs-views", "6", "--allpairs-desc", "2048"]
    two = _run(args, {"SFM_BENCH_BACKEND": "gloo"}, 600)
    assert two["n_gpus"] == 2 and two["config"]["pairs_per_step"] == 16 and two["value"] > 0
    ap2 = two["allpairs"]
    assert ap2["n_gpus"] == 2 and ap2["pairs"] == 15 and ap2["gathered_bytes_per_rank"] > 0
    one = _run(["--gpus", "1"] + args[2:], {}, 600)
    ap1 = one["allpairs"]
    assert one["n_gpus"] == 1 and ap1["gathered_bytes_per_rank"] == 0
    # the sharded run finds what one process finds
    assert ap1["inliers_neighbouring_views_min_max"] == ap2["inliers_neighbouring_views_min_max"]
    assert ap1["inliers_farthest_pair"] == ap2["inliers_farthest_pair"]
