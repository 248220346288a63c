"""Builds libsfm_gpu.so (HIP, gfx950 only) in-tree with hipcc, and beside it
libsfm_gpu_timing.so: the same sources with -DSFM_GPU_TIMING_BUILD, which adds
the timing ablations of the kernels (parts switched off, wrong results).  Only
bench.py's ceiling measurement and the profiling scripts load the timing build;
a host links libsfm_gpu.so, which holds no switch that changes a result."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsfm_gpu.so")
LIB_TIMING = os.path.join(HERE, "libsfm_gpu_timing.so")
SOURCES = ["sfm_gpu_api.hip", "knn_kernels.hip", "ransac_kernels.hip", "pair_kernels.hip",
           "ransac_api.hip", "pose.hip", "tracks.hip", "pnp.hip"]
# -ffp-contract=off: the RANSAC kernels must round exactly like unfused IEEE
# double arithmetic so that inlier masks can be compared bit-for-bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, jobs=8):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "sfm_gpu.h"))
    headers.append(os.path.join(ROOT, "include", "sfm_gpu_debug.h"))
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    targets = [(LIB, ".o", []), (LIB_TIMING, ".timing.o", ["-DSFM_GPU_TIMING_BUILD"])]
    todo = []
    objs = {lib: [] for lib, _, _ in targets}
    for lib, suffix, extra in targets:
        for src in sources:
            s = os.path.join(CSRC, src)
            o = os.path.join(CSRC, src.replace(".hip", suffix))
            objs[lib].append(o)
            if force or _stale(o, [s] + headers):
                todo.append((src, [hipcc] + FLAGS + extra + ["-c", s, "-o", o]))
    running = []
    failed = []

    def reap(block):
        for item in list(running):
            src, p = item
            if block:
                p.wait()
            if p.poll() is not None:
                running.remove(item)
                if p.returncode != 0:
                    failed.append(src)
                if block:
                    return

    for src, cmd in todo:
        while len(running) >= jobs:
            reap(True)
        if verbose:
            print(" ".join(cmd))
        running.append((src, subprocess.Popen(cmd)))
    while running:
        reap(True)
    if failed:
        raise RuntimeError("hipcc failed on %s" % ", ".join(failed))
    for lib, _, _ in targets:
        if force or todo or _stale(lib, objs[lib]):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", lib] + objs[lib]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
