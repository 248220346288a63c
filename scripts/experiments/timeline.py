"""Kernel timeline of a few consecutive steps from a rocprofv3 kernel trace of bench.py.
timeline.py <..._kernel_trace.csv> <index of the first sweep launch> [number of steps]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:30]


idx = [i for i, r in enumerate(rows) if "knn2_mfma16" in r["Kernel_Name"]]
print("%d sweep launches in the trace" % len(idx))
i0, i1 = idx[first], idx[first + steps]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-28s q%-3s start %8.1f  dur %7.1f  end %8.1f" % (short(r["Kernel_Name"]), r.get("Queue_Id", "?"), s / 1e3, (e - s) / 1e3, e / 1e3))
