#!/bin/bash
# A/B of the sweep's workgroup shapes on the bench workload (4: four waves, 5: eight waves)
F="--no-cpu-baseline --no-microkernel --no-overlap-probe --no-configs2 --no-allpairs --steps 100 --warmup 10"
for i in 1 2; do
for v in ${VARIANTS:-4 5}; do
timeout -k 10 120 python bench.py $F --knn-variant $v > gpurun_out/ab_v${v}_$i.json 2>gpurun_out/ab_v${v}_$i.err || { tail -5 gpurun_out/ab_v${v}_$i.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_v${v}_$i.json").read().strip().splitlines()[-1])
print("variant $v", d["value"], d["ms_per_step"], "sweep", d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
PY
done
done
