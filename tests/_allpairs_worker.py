"""Worker of test_allpairs_gloo.py: one rank of a gloo job (or a single
process when WORLD_SIZE is 1) running sfm_amd.allpairs.match_all_pairs with the
CPU oracle as the arithmetic backend.  Usage: _allpairs_worker.py OUT.npz MODE"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from oracle import pyoracle as po
from sfm_amd import allpairs, synth

SIZES = [192, 256, 160, 224, 208]  # rows per image
N_HYP = 48


def make_image(g):
    """Image g of a small synthetic collection in which consecutive images share planted matches."""
    n = SIZES[g]
    rng = np.random.default_rng(500 + g)
    base = synth.sift_like(256, 77)           # common pool: images overlap through it
    take = rng.permutation(256)[:n]
    desc = base[take].copy()
    noise = rng.integers(-3, 4, size=desc.shape)
    desc = np.clip(desc + noise, 0, 255).astype(np.float32)
    kp = rng.uniform(0, 1000, size=(n, 2)).astype(np.float32)
    return desc, kp


class OracleBackend:
    """Blob = descriptors (float32) followed by keypoints (float32), zero padded."""

    @staticmethod
    def rows_of(image):
        return int(image[0].shape[0])

    @staticmethod
    def blob_bytes(max_rows):
        return max_rows * (128 + 2) * 4

    def prepare(self, images, max_rows):
        blobs = torch.zeros((len(images), self.blob_bytes(max_rows)), dtype=torch.uint8)
        rows = torch.tensor([self.rows_of(im) for im in images], dtype=torch.int32)
        for k, (desc, kp) in enumerate(images):
            raw = np.concatenate([desc.astype(np.float32).ravel(), kp.astype(np.float32).ravel()]).view(np.uint8)
            blobs[k, : raw.size] = torch.from_numpy(raw.copy())
        return blobs, rows

    @staticmethod
    def unpack(blob, n):
        f = blob.numpy().view(np.float32)
        return f[: n * 128].reshape(n, 128), f[n * 128: n * 130].reshape(n, 2)

    def match(self, blobs, rows, pairs):
        ng, ni = [], []
        for a, b in pairs:
            da, ka = self.unpack(blobs[int(a)], int(rows[a]))
            db, kb = self.unpack(blobs[int(b)], int(rows[b]))
            m, c = po.knn_match(da, db, 2)
            keep = po.ratio_test(m, c, 0.7)
            good = m[keep, 0]
            p1 = ka[good["queryIdx"]].astype(np.float64)
            p2 = kb[good["trainIdx"]].astype(np.float64)
            r = po.ransac(p1, p2, None, model=po.MODEL_FUNDAMENTAL, prob=0.99, threshold=3.0,
                          fixed_count=N_HYP)
            ng.append(len(good))
            ni.append(int(r["n_inliers"]) if r["ok"] else 0)
        return np.array(ng), np.array(ni)


def main():
    out, mode = sys.argv[1], sys.argv[2]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    V = len(SIZES)
    lo, hi = allpairs.owned_range(V, world, rank)
    local = [make_image(g) for g in range(lo, hi)]
    pairs, good, inl = allpairs.match_all_pairs(local, V, OracleBackend(), mode=mode)
    np.savez(out, pairs=pairs, good=good, inl=inl)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
