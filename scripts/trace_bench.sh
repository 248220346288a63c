#!/bin/bash
# Kernel-trace statistics of `python3 bench.py` only (no counter passes).
# Usage: scripts/trace_bench.sh <tag> [extra bench.py flags]
set -o pipefail
TAG=${1:-t}
shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- python3 $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-microkernel --no-overlap-probe --no-records-to-host --no-configs2 --no-allpairs "$@" > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
cp "$f" "$OUT/kernel_stats.csv"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_\w+)", r["Name"])
    print("%-46s calls %5s avg %9.1f us total %6.2f %%" % ((m.group(1) if m else r["Name"][:40]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
