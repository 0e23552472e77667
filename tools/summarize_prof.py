#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel stats + per-kernel mean of each PMC counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    for p in ("void ", "hmj::"):
        name = name.replace(p, "")
    return name[:60]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("# kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("%-62s calls=%-5s total_ms=%10.3f avg_us=%10.2f pct=%s" % (
            short(row.get("Name", "")), row.get("Calls"), float(row.get("TotalDurationNs", 0)) / 1e6,
            float(row.get("AverageNs", 0)) / 1e3, row.get("Percentage")))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("# pmc:", os.path.basename(d))
        for k, cs in sorted(acc.items()):
            if "gen_" in k or "elementwise" in k:
                continue
            print("  %-58s %s" % (k, "  ".join("%s=%.4g(n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items()))))
