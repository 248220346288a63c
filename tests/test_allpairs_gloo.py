"""Multi-process path of the pair matcher (sfm_amd/allpairs.py) on CPU: two gloo
ranks must produce exactly what one process produces.  The arithmetic backend
is the CPU oracle (test infrastructure); the GPU backend is exercised by
tests/test_gpu_pairs.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from sfm_amd import allpairs

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_allpairs_worker.py")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, out_dir, mode):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        out = os.path.join(out_dir, "w%d_r%d_%s.npz" % (world, r, mode))
        procs.append((subprocess.Popen([sys.executable, WORKER, out, mode], env=env), out))
    res = []
    for p, out in procs:
        assert p.wait(timeout=300) == 0
        res.append(np.load(out))
    return res


def test_pair_list_and_ownership():
    ring = allpairs.pair_list(4, "ring")
    assert ring.tolist() == [[0, 1], [1, 2], [2, 3], [3, 0]]  # main.cpp:167-173
    allp = allpairs.pair_list(4, "all")
    assert len(allp) == 6 and all(i < j for i, j in allp)
    assert allpairs.pair_list(1, "ring").shape == (0, 2)
    cover = []
    for r in range(3):
        lo, hi = allpairs.owned_range(7, 3, r)
        cover += list(range(lo, hi))
    assert cover == list(range(7))


def test_deal_pairs_is_a_partition_and_balanced():
    rows = np.array([100, 4000, 8000, 8000, 200, 6000])
    pairs = allpairs.pair_list(6, "all")
    parts = allpairs.deal_pairs(pairs, rows, 4)
    got = np.sort(np.concatenate(parts))
    assert got.tolist() == list(range(len(pairs)))
    cost = rows[pairs[:, 0]] * rows[pairs[:, 1]]
    loads = np.array([cost[p].sum() for p in parts])
    assert loads.max() <= cost.sum() / 4 + cost.max()
    # same answer on every call (ranks compute it independently)
    again = allpairs.deal_pairs(pairs, rows, 4)
    assert all((a == b).all() for a, b in zip(parts, again))
    empty = allpairs.deal_pairs(np.zeros((0, 2), int), rows, 2)
    assert [len(e) for e in empty] == [0, 0]


@pytest.mark.parametrize("mode", ["all", "ring"])
def test_two_gloo_ranks_equal_one_process(tmp_path, mode):
    one = run_world(1, str(tmp_path), mode)[0]
    two = run_world(2, str(tmp_path), mode)
    for r in two:
        assert (r["pairs"] == one["pairs"]).all()
        assert (r["good"] == one["good"]).all()
        assert (r["inl"] == one["inl"]).all()
    # the workload is not trivial: overlapping images do match
    assert one["good"].sum() > 0 and one["inl"].sum() > 0


@pytest.mark.parametrize("world", [3, 4, 8])
def test_more_ranks_than_divide_the_images_equal_one_process(tmp_path, world):
    """Five images over 3, 4 and 8 ranks: uneven ownership, and with 8 ranks three
    of them own no image at all and still take part in the all-gather, the deal
    and the all-reduce."""
    one = run_world(1, str(tmp_path), "all")[0]
    many = run_world(world, str(tmp_path), "all")
    assert len(many) == world
    for r in many:
        assert (r["pairs"] == one["pairs"]).all()
        assert (r["good"] == one["good"]).all()
        assert (r["inl"] == one["inl"]).all()
