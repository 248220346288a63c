"""Dispatch sequence of one step of bench.py from a rocprofv3 kernel trace:
every launch from one distance sweep to the next, with its duration, the gap
before it (negative: it started beside the previous launches) and its queue.
    python scripts/step_sequence.py <kernel_trace.csv> [index of the sweep, default 5] [how many steps, default 1]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sweeps = [i for i, r in enumerate(rows) if "knn2_mfma" in r["Kernel_Name"]]
if len(sweeps) <= first + count:
    raise SystemExit("not enough steps in the trace")
starts = [int(rows[i]["Start_Timestamp"]) for i in sweeps]
print("sweep start to sweep start (us):", [round((b - a) / 1e3) for a, b in zip(starts, starts[1:])])
lo, hi = sweeps[first], sweeps[first + count]
prev_end = None
for r in rows[lo:hi + 1]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:50]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    print("%-44s %8.1f us  gap %7.1f  q%s grid %s wg %s" % (name, (en - st) / 1e3, gap, r.get("Queue_Id", "?"), grid,
                                                          r.get("Workgroup_Size_X", "?")))
    prev_end = max(prev_end or 0, en)
