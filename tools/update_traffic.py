#!/usr/bin/env python3
"""profiles/traffic_latest.json (read by bench.py for roofline.traffic) from a tools/profile_bench.sh
traffic summary.  usage: tools/update_traffic.py SUMMARY_traffic.json WORKLOAD KERNEL_SUBSTRING
The entry records the hash of the library sources it was measured on (bench.py's csrc_sha16): bench.py reports the
figure only while the sources are unchanged."""
import glob
import hashlib
import json
import os
import sys

summary, workload, sub = sys.argv[1:4]
t = json.load(open(summary))
name = [k for k in t["kernels"] if sub in k][0]
v = t["kernels"][name]
out = {"workload": workload, "kernel": name, "hbm_bytes_per_launch": v["hbm_bytes"],
       "read_bytes": v["hbm_bytes"] - v["WRITE_SIZE_bytes"], "write_bytes": v["WRITE_SIZE_bytes"],
       "FETCH_SIZE_bytes_raw": v["FETCH_SIZE_bytes_raw"], "rdreq": v.get("rdreq"),
       "profile": os.path.relpath(os.path.abspath(summary), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
       "source": "rocprofv3 --pmc, separate passes for FETCH_SIZE, WRITE_SIZE "
                 "and TCC_EA0_RDREQ by size; reads = 128-B requests x 128 = 2 x FETCH_SIZE on gfx950; "
                 "tools/profile_bench.sh, tools/profile_summary.py)"}
# all kernels of a vmult (cell loop launches + pass 2): the library's kernels launched at least as often as the
# dominant one (setup kernels run once), per vmult = per launch of the dominant kernel
per_vmult = {k: w["hbm_bytes"] * w["launches"] / v["launches"] for k, w in t["kernels"].items()
             if k.startswith("mfgpu::") and w["launches"] >= v["launches"]}
out["step_hbm_bytes_per_vmult"] = sum(per_vmult.values())
out["step_kernels"] = per_vmult
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hsh = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "dealii-cuda_amd", "csrc", "*")) + [os.path.join(root, "include", "mfgpu.h")]):
    hsh.update(open(f, "rb").read())
out["csrc_sha16"] = hsh.hexdigest()[:16]
json.dump(out, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1)
print(name, out["hbm_bytes_per_launch"])
