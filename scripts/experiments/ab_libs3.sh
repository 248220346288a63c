#!/bin/bash
# A/B of two library builds on bench.py's headline and records-to-host loops (default step counts): ab_libs3.sh A B [rounds]
A=$1; B=$2; N=${3:-3}
F="--no-cpu-baseline --no-microkernel --no-overlap-probe --no-configs2 --no-allpairs"
for i in $(seq $N); do
for L in $A $B; do
timeout -k 10 120 python scripts/experiments/bench_with_lib.py $L $F > gpurun_out/ab_lib.json 2> gpurun_out/ab_lib.err || { tail -5 gpurun_out/ab_lib.err; exit 1; }
python - $L <<'PY'
import json, sys, os
d = json.loads(open("gpurun_out/ab_lib.json").read().strip().splitlines()[-1])
t = d["with_records_to_host"]
print("%-28s %.4f ms  sweep %.1f  to host %.4f ms  host copy ok %s" % (os.path.basename(sys.argv[1]), d["ms_per_step"], d["roofline"]["avg_launch_us"], t["ms_per_step"], t["host_copy_equals_device_records"]))
PY
done
done
