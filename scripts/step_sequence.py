"""Dispatch sequence of one step of bench.py from a rocprofv3 kernel trace:
every launch between two descriptor conversions, with its duration and the gap
before it.
    python scripts/step_sequence.py <kernel_trace.csv> [step index from the end, default 1]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# a step on the main stream starts at its upload kernel
starts = [i for i, r in enumerate(rows) if "upload_block" in r["Kernel_Name"]]
if len(starts) <= back:
    raise SystemExit("not enough steps in the trace")
lo, hi = starts[-1 - back], starts[-back]
prev_end = None
for r in rows[lo:hi]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:50]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    print("%-52s %8.1f us  gap %6.1f  grid %s" % (name, (en - st) / 1e3, gap, grid))
    prev_end = max(prev_end or 0, en)
print("step: %.1f us from first start to last end" % ((prev_end - int(rows[lo]["Start_Timestamp"])) / 1e3))
