#!/bin/bash
# Kernel statistics of the exhaustive matcher on one GPU (scripts/allpairs_perf.py 64 20000 E)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_allpairs
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 $REPO/scripts/allpairs_perf.py 64 20000 E > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
grep -v "rocprofv3\|^W2\|^E2" "$OUT/trace.log" | tail -8
python3 - "$OUT/trace/t_kernel_stats.csv" <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:18]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_\w+)", r["Name"])
    print("%-44s calls %5s avg %9.1f us total %8.2f ms %6.2f %%" % ((m.group(1) if m else r["Name"][:40]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
