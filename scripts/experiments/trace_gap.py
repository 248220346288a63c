"""From a rocprofv3 kernel trace (csv) of bench.py: the time from the end of each distance sweep to the start of
the ratio compaction behind it, and what ran in between.  trace_gap.py <..._kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:30]


gaps = []
between = defaultdict(list)
i = 0
while i < len(rows):
    if "knn2_mfma16" in rows[i]["Kernel_Name"]:
        end = int(rows[i]["End_Timestamp"])
        j = i + 1
        while j < len(rows) and "ratio_compact" not in rows[j]["Kernel_Name"]:
            between[short(rows[j]["Kernel_Name"])].append(int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"]))
            j += 1
        if j < len(rows):
            gaps.append(int(rows[j]["Start_Timestamp"]) - end)
        i = j
    else:
        i += 1
gaps = gaps[len(gaps) // 4:]
print("sweep end -> compaction start: mean %.1f us over %d steps" % (sum(gaps) / len(gaps) / 1e3, len(gaps)))
for k, v in between.items():
    print("   %-28s %5d launches, mean %.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
