#!/usr/bin/env python3
"""profiles/traffic_latest.json (read by bench.py for roofline.traffic) from a tools/profile_bench.sh
traffic summary.  usage: tools/update_traffic.py SUMMARY_traffic.json WORKLOAD KERNEL_SUBSTRING"""
import json
import os
import sys

summary, workload, sub = sys.argv[1:4]
t = json.load(open(summary))
name = [k for k in t["kernels"] if sub in k][0]
v = t["kernels"][name]
out = {"workload": workload, "kernel": name, "hbm_bytes_per_launch": v["hbm_bytes"],
       "read_bytes_calibrated": v["read_bytes_calibrated"], "write_bytes": v["WRITE_SIZE_bytes"],
       "fetch_calibration": t["fetch_calibration"],
       "source": "profiles/r01_bench_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                 "passes, tools/profile_bench.sh)"}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1)
print(name, out["hbm_bytes_per_launch"])
