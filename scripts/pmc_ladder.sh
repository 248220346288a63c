#!/bin/bash
# One PMC pass over scripts/sweep_ladder.py (timing build): per sweep variant
# (template arguments in the kernel name) the per-launch mean of each counter.
# Usage: scripts/pmc_ladder.sh <tag> <counter> [<counter> ...]
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcl_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --pmc "$@" -d "$OUT/run" -o lad --output-format csv -- python3 $REPO/scripts/sweep_ladder.py 1 > "$OUT/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/pmc.log"; exit 1; }
f=$(find "$OUT/run" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$OUT/summary.json" <<'PY'
import csv, sys, re, json
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
disp = defaultdict(dict)
for r in rows:
    m = re.search(r"knn2_mfma16_kernel<(\d+), ?(\d+)>", r["Kernel_Name"])
    if not m:
        continue
    key = "waves%s_mask%s_grid%s" % (m.group(1), m.group(2), r.get("Grid_Size", r.get("Grid_Size_X")))
    d = disp[(key, r["Dispatch_Id"])]
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for (key, _), d in disp.items():
    for c, v in d.items():
        a = acc[key][c]
        a[0] += v
        a[1] += 1
res = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in acc.items()}
json.dump(res, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k in sorted(res):
    print(k, {c: int(v) for c, v in res[k].items()})
PY
