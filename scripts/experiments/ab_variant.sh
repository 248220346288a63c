#!/bin/bash
# Sweep shapes against each other on the bench workload, interleaved: ab_variant.sh "4 5" [rounds]
VS=${1:-"4 5"}; N=${2:-3}
F="--no-cpu-baseline --no-microkernel --no-overlap-probe --no-records-to-host --no-configs2 --no-allpairs"
for i in $(seq $N); do
for v in $VS; do
timeout -k 10 120 python bench.py $F --knn-variant $v > gpurun_out/ab_var.json 2> gpurun_out/ab_var.err || { tail -5 gpurun_out/ab_var.err; exit 1; }
python - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_var.json").read().strip().splitlines()[-1])
print("variant %s  %.4f ms  sweep %.1f us  frac %.4f" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
done
