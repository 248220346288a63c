#!/bin/bash
# rocTX ranges of the library's stages beside the kernel trace (SFM_GPU_ROCTX=1)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_markers
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SFM_GPU_ROCTX=1
timeout -k 10 240 rocprofv3 --marker-trace --kernel-trace -d "$OUT/trace" -o t --output-format csv -- python3 $REPO/scripts/pairs_perf.py F fixed 8 2048 64 > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
ls "$OUT/trace"
head -3 "$OUT/trace/t_marker_api_trace.csv"
cut -d, -f2 "$OUT/trace/t_marker_api_trace.csv" | sort | uniq -c | sort -rn | head -12
