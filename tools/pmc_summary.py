#!/usr/bin/env python3
"""Averages rocprofv3 counter_collection.csv rows per kernel: tools/pmc_summary.py DIR [kernel-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "apply_batches"
for f in sorted(glob.glob(f"{d}/*/*/*counter_collection.csv")):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    print(f.split("/")[-3], {k: round(v / cnt[k], 1) for k, v in agg.items()}, "launches", max(cnt.values()) if cnt else 0)
