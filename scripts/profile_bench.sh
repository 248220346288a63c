#!/bin/bash
# Profiles `python3 bench.py` on the GPU box: kernel-trace statistics, then PMC
# passes (one counter group per run).  Results land in gpurun_out/prof_<tag>/.
# Usage: scripts/profile_bench.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microkernel --no-overlap-probe --no-records-to-host --no-configs2 --no-allpairs"
# the trace pass at bench.py's own step counts, so that its per-kernel averages are the steady state the bench line reports
TRACE_BENCH="python3 $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-microkernel --no-overlap-probe --no-records-to-host --no-configs2 --no-allpairs"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- $TRACE_BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
echo "trace done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $pass -d "$OUT/pmc_$name" -o bench --output-format csv -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
  echo "pmc $name done"
done
python3 $REPO/scripts/summarise_profile.py "$OUT"
