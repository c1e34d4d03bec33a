#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh run into small files fit for profiles/:
   <out>_kernel_stats.csv  (rocprofv3 --stats per-kernel summary)
   <out>_traffic.json      (HBM bytes per launch of the dominant kernel from the PMC passes)

HBM-traffic recipe (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x and other widths are uncalibrated, so the read
side is CALIBRATED in the same run on a kernel of known traffic with the same 8-byte-per-lane access
shape: fold_kernel reads two coalesced arrays of n_cells*nd doubles and writes one."""
import collections
import csv
import glob
import json
import shutil
import sys

d, out = sys.argv[1], sys.argv[2]


def per_kernel(sub, counter):
    f = glob.glob(f"{d}/{sub}/*/*counter_collection.csv")
    if not f:
        return {}
    agg, cnt, grid = collections.defaultdict(float), collections.Counter(), {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k] += float(r["Counter_Value"])
        cnt[k] += 1
        grid[k] = int(r["Grid_Size"])
    return {k: (agg[k] / cnt[k], cnt[k], grid[k]) for k in agg}


ks = glob.glob(f"{d}/kt/*/*kernel_stats.csv")
if ks:
    shutil.copy(ks[0], out + "_kernel_stats.csv")
fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
hit, miss = per_kernel("l2", "TCC_HIT_sum"), per_kernel("l2", "TCC_MISS_sum")
res = {"units": "bytes per launch", "kernels": {}}
cal = None
for k in fetch:
    if "fold_kernel" in k:
        n_elem = fetch[k][2]  # grid size = threads >= n_cells*nd, rounded to 256
        known_read = 2 * 8 * n_elem
        cal = known_read / (fetch[k][0] * 1024)
        res["fetch_calibration"] = {"kernel": k, "known_read_bytes": known_read, "FETCH_SIZE_KiB": fetch[k][0],
                                    "factor": cal, "known_write_bytes": 8 * n_elem,
                                    "WRITE_SIZE_KiB": write.get(k, (0,))[0]}
for k in fetch:
    rd = fetch[k][0] * 1024
    wr = write.get(k, (0, 0, 0))[0] * 1024
    res["kernels"][k] = {"launches": fetch[k][1], "FETCH_SIZE_bytes_raw": rd, "WRITE_SIZE_bytes": wr,
                         "read_bytes_calibrated": rd * cal if cal else None,
                         "hbm_bytes": (rd * cal if cal else rd * 2) + wr,
                         "l2_hit_rate": (hit[k][0] / (hit[k][0] + miss[k][0])) if k in hit and hit[k][0] + miss[k][0] > 0 else None}
json.dump(res, open(out + "_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
