"""Per-step GPU busy / idle time from a rocprofv3 kernel trace of bench.py
(steps are cut at the descriptor conversion kernel).
    python scripts/step_gaps.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    if "prep_rows" in r["Kernel_Name"] and cur:
        steps.append(cur)
        cur = []
    cur.append(r)
steps.append(cur)
prev_start = None
for i, s in enumerate(steps):
    st, en = int(s[0]["Start_Timestamp"]), int(s[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s)
    big = []
    for a, b in zip(s, s[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if g > 3:
            m = re.search(r"(\w+_kernel|__amd_\w+)", a["Kernel_Name"])
            big.append("%.0f after %s" % (g, m.group(1) if m else "?"))
    print("step %d: %d launches, start-to-start %.1f us, busy %.1f us; gaps > 3 us: %s" %
          (i, len(s), (st - prev_start) / 1e3 if prev_start else 0.0, busy / 1e3, ", ".join(big)))
    prev_start = st
