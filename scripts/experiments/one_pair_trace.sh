#!/bin/bash
# Kernel trace of one 32768 x 32768 pair with essential RANSAC, 4096 hypotheses (scripts/pairs_perf.py E fixed 1 32768 4096)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_onepair
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 $REPO/scripts/pairs_perf.py E fixed 1 32768 4096 > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
tail -12 "$OUT/trace.log"
python3 - "$OUT/trace/t_kernel_trace.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [i for i, r in enumerate(rows) if "ransac_select" in r["Kernel_Name"]]
lo = sel[-2] + 1
prev = None
for r in rows[lo:sel[-1] + 1]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s %8.1f us gap %6.1f grid %sx%sx%s" % (m.group(1) if m else r["Kernel_Name"][:40], (en - st) / 1e3, (st - prev) / 1e3 if prev else 0, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
    prev = en
PY
