#!/bin/bash
# As ab_libs.sh, printing also the records-to-host loop's figure:  ab_libs2.sh <lib A> <lib B> [rounds]
A=$1; B=$2; N=${3:-3}
F="--no-cpu-baseline --no-microkernel --no-overlap-probe --no-configs2 --no-allpairs --steps 100 --warmup 10"
for i in $(seq $N); do
for L in $A $B; do
timeout -k 10 120 python scripts/experiments/bench_with_lib.py $L $F > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err || { tail -5 gpurun_out/ab_lib.err; exit 1; }
python - $L <<'PY'
import json, sys, os
d = json.loads(open("gpurun_out/ab_lib.json").read().strip().splitlines()[-1])
print("%-28s %.4f ms  sweep %.1f  to host %.4f ms" % (os.path.basename(sys.argv[1]), d["ms_per_step"], d["roofline"]["avg_launch_us"], d["with_records_to_host"]["ms_per_step"]))
PY
done
done
