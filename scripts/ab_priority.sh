F="--no-cpu-baseline --no-microkernel --no-overlap-probe --no-configs2 --no-allpairs --steps 100 --warmup 10"
for i in 1 2; do
for pr in 0 -1; do
timeout -k 10 120 python bench.py $F --main-priority $pr > gpurun_out/ab_p${pr}_$i.json 2>/dev/null || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_p${pr}_$i.json").read().strip().splitlines()[-1])
print("prio $pr", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
done
done
