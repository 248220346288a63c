"""Capture-order recovery from the match-count matrix (host logic)."""
import numpy as np
import pytest

from sfm_amd import allpairs


def _banded(V, rng, width=3):
    c = np.zeros((V, V), np.int64)
    for i in range(V):
        for d in range(1, width + 1):
            if i + d < V:
                c[i, i + d] = int(3000 / d ** 2 + rng.integers(0, 50))
    return c


def _weight(c, parent):
    w = c + c.T
    return sum(int(w[v, p]) for v, p in enumerate(parent) if p >= 0)


def test_sequence_shuffled_is_chained_back():
    rng = np.random.default_rng(1)
    V = 40
    c = _banded(V, rng)
    perm = rng.permutation(V)                   # image number -> position in the sequence
    shuffled = np.zeros_like(c)
    for i in range(V):
        for j in range(V):
            if c[i, j]:
                a, b = np.where(perm == i)[0][0], np.where(perm == j)[0][0]
                shuffled[min(a, b), max(a, b)] = c[i, j]      # upper triangle, as the matcher fills it
    order, parent = allpairs.spanning_order(shuffled)
    assert sorted(order.tolist()) == list(range(V))
    assert (parent == -1).sum() == 1 and parent[order[0]] == -1
    # every image hangs on a neighbour in the true sequence
    for v, p in enumerate(parent):
        if p >= 0:
            assert abs(int(perm[v]) - int(perm[p])) == 1
    # a parent is always visited before its child
    pos = np.argsort(order)
    assert all(pos[p] < pos[v] for v, p in enumerate(parent) if p >= 0)


def test_is_a_maximum_spanning_tree():
    rng = np.random.default_rng(2)
    for V in (2, 5, 9):
        c = np.triu(rng.integers(0, 100, (V, V)), 1)
        order, parent = allpairs.spanning_order(c)
        # exhaustive check against Kruskal on the same weights
        w = c + c.T
        edges = sorted(((int(w[i, j]), i, j) for i in range(V) for j in range(i + 1, V) if w[i, j] > 0), reverse=True)
        comp = list(range(V))

        def find(x):
            while comp[x] != x:
                x = comp[x]
            return x
        best = 0
        for wt, i, j in edges:
            a, b = find(i), find(j)
            if a != b:
                comp[a] = b
                best += wt
        assert _weight(c, parent) == best


def test_disconnected_groups_and_ties_are_deterministic():
    c = np.zeros((6, 6), np.int64)
    c[0, 1] = c[1, 2] = 7                       # one group, equal weights
    c[4, 5] = 9                                 # another
    order, parent = allpairs.spanning_order(c)  # image 3 matches nothing
    assert order.tolist() == [1, 0, 2, 4, 5, 3]
    assert parent.tolist() == [1, -1, 1, -1, -1, 4]
    again = allpairs.spanning_order(c)
    assert (again[0] == order).all() and (again[1] == parent).all()
    with pytest.raises(ValueError):
        allpairs.spanning_order(np.zeros((2, 3)))
    o, p = allpairs.spanning_order(np.zeros((0, 0)))
    assert len(o) == 0 and len(p) == 0
