"""Cost of the exhaustive float-ranked pass: kNN of one image pair whose queries ALL need it (every row of both
images the same constant, so every group minimum ties), a few (rows duplicated for a fraction of the queries) or
none.  exhaustive_cost.py [path to another libsfm_gpu.so]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from sfm_amd import _lib

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from sfm_amd import matcher, synth

torch.cuda.init()
ctx = matcher.Context(0)
n = 8192
Q, T, _ = synth.planted_pair(n, n, 1001, 1002, 1003)
cases = {"sift-like": (Q, T)}
c = np.full((n, 128), 37.0, np.float32)
cases["all rows equal"] = (c, c.copy())
T2 = T.copy()
T2[1::2] = T2[0::2]  # every train row twice: first and second neighbour tie for the planted queries
cases["train rows in equal pairs"] = (Q, T2)
out = torch.empty(n * 2 * 16, dtype=torch.uint8, device="cuda")
cnt = torch.empty(n, dtype=torch.int32, device="cuda")
for name, (q, t) in cases.items():
    iq = matcher.Image(ctx, q)
    it = matcher.Image(ctx, t)
    for _ in range(3):
        matcher.knn_match_images(ctx, iq, it, 2, (out.data_ptr(), cnt.data_ptr()))
    ctx.synchronize()
    rc = ctypes.c_int32(0)
    ctx.check(ctx._L.sfm_gpu_debug_redo_count(ctx.handle, ctypes.byref(rc)))
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        matcher.knn_match_images(ctx, iq, it, 2, (out.data_ptr(), cnt.data_ptr()))
    ctx.synchronize()
    print("%-28s queries through the exhaustive pass %5d   %.1f us per call" % (name, rc.value, (time.perf_counter() - t0) / reps * 1e6), flush=True)
    iq.close()
    it.close()
