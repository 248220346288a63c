#!/bin/bash
# A/B of the scoring schedules on the bench workload (prune mode 1 = default, 2 = earlier two-cut schedule).
F="--no-cpu-baseline --no-microkernel --no-overlap-probe --no-configs2 --no-allpairs --steps 100 --warmup 10"
for i in 1 2; do
for pm in ${MODES:-1 2}; do
timeout -k 10 120 python bench.py $F --score-prune $pm > gpurun_out/ab_pm${pm}_$i.json 2>gpurun_out/ab_pm${pm}_$i.err || { tail -5 gpurun_out/ab_pm${pm}_$i.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_pm${pm}_$i.json").read().strip().splitlines()[-1])
k=d["kernels_outside_timed_region"]
print("prune $pm", d["value"], d["ms_per_step"], "sweep", d["roofline"]["avg_launch_us"], "score", k["ransac_score"]["avg_us"], "inl", d["config"]["n_inliers_mean"])
PY
done
done
