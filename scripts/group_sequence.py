"""Dispatch sequence between two consecutive distance sweeps of a rocprofv3 kernel
trace (one launch group of sfm_gpu_match_pairs): every launch with its duration
and the idle gap before it, then busy / idle totals.
    python scripts/group_sequence.py <kernel_trace.csv> [group index from the end, default 2]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
starts = [i for i, r in enumerate(rows) if "knn2_mfma" in r["Kernel_Name"]]
lo, hi = starts[-1 - back], starts[-back]
prev_end = None
busy = idle = 0.0
for r in rows[lo:hi]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_\w+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:50]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print("%-52s %8.1f us  gap %6.1f" % (name, (en - st) / 1e3, gap))
    busy += (en - st) / 1e3
    idle += max(gap, 0.0)
    prev_end = max(prev_end or 0, en)
print("group: busy %.1f us, idle %.1f us" % (busy, idle))
