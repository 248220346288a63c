#!/bin/bash
# One PMC pass over `python3 bench.py`, printing per-dispatch counters for the
# kernels whose name matches a pattern.
# Usage: scripts/pmc_bench.sh <tag> <kernel regex> <counter> [<counter> ...]
set -o pipefail
TAG=$1; PAT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" -d "$OUT/run" -o bench --output-format csv -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-microkernel --no-configs2 --no-allpairs > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
f=$(find "$OUT/run" -name '*counter_collection.csv' | head -1)
cp "$f" "$OUT/counters.csv"
python3 - "$OUT/counters.csv" "$PAT" <<'PY'
import csv, sys, re
from collections import OrderedDict
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
disp = OrderedDict()
for r in rows:
    if not pat.search(r["Kernel_Name"]):
        continue
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": re.search(r"\w+_kernel", r["Kernel_Name"]).group(0), "grid": (r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"))})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in list(disp)[-8:]:
    d = disp[k]
    print(k, d["name"], d["grid"], {c: int(v) for c, v in d.items() if c not in ("name", "grid")})
PY
