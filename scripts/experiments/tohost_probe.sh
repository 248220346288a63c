#!/bin/bash
# The records-to-host figure of bench.py under different flag sets (which block before it matters?).
run() {
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-configs2 --no-allpairs "$@" > gpurun_out/th.json 2> gpurun_out/th.err || { tail -5 gpurun_out/th.err; exit 1; }
  python - "$*" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/th.json").read().strip().splitlines()[-1])
print("%-60s %.4f ms  to host %.4f ms" % (sys.argv[1], d["ms_per_step"], d["with_records_to_host"]["ms_per_step"]))
PY
}
run
run --steps 100
run --no-overlap-probe
run --no-microkernel
run --no-overlap-probe --no-microkernel
run --steps 100 --no-overlap-probe --no-microkernel
