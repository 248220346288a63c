#!/bin/bash
# A/B of two builds of the library on the bench workload, interleaved:  ab_libs.sh <lib A> <lib B> [rounds]
# Prints ms per step, the sweep's launch duration and the solver's and scoring's per-step figures.
A=$1; B=$2; N=${3:-3}
F="--no-cpu-baseline --no-microkernel --no-overlap-probe --no-configs2 --no-allpairs --steps 100 --warmup 10"
for i in $(seq $N); do
for L in $A $B; do
timeout -k 10 120 python scripts/experiments/bench_with_lib.py $L $F > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err || { tail -5 gpurun_out/ab_lib.err; exit 1; }
python - $L <<'PY'
import json, sys, os
d = json.loads(open("gpurun_out/ab_lib.json").read().strip().splitlines()[-1])
k = d["kernels_outside_timed_region"]
print("%-28s %.4f ms  sweep %.1f  solve %.1f  score %.1f" % (os.path.basename(sys.argv[1]), d["ms_per_step"], d["roofline"]["avg_launch_us"], k["ransac_solve"]["avg_us"], k["ransac_score"]["avg_us"]))
PY
done
done
