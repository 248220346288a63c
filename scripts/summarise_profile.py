#!/usr/bin/env python3
"""Condenses a scripts/profile_bench.sh output directory into small files:
kernel_stats.csv (copy of rocprofv3's per-kernel statistics) and pmc.json
(per kernel: launches and per-launch mean of every counter collected)."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd_\w+)", name)
    return m.group(1) if m else name[:80]


def main(out):
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    res = {k: {c: {"mean_per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()}
           for k, cs in acc.items()}
    json.dump(res, open(os.path.join(out, "pmc.json"), "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        print(k, {c: round(v["mean_per_launch"], 1) for c, v in res[k].items()})


if __name__ == "__main__":
    main(sys.argv[1])
