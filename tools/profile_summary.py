#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh run into small files fit for profiles/:
   <out>_kernel_stats.csv  (rocprofv3 --stats per-kernel summary)
   <out>_traffic.json      (HBM bytes per launch of the dominant kernel from the PMC passes)

HBM-traffic recipe (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate --pmc passes.  On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B although the requests are 128 B
wide, i.e. it reports HALF of the bytes read ("double it").  A third pass collects the L2's fabric read
requests by size (TCC_EA0_RDREQ_{32B,64B,128B}); for these kernels every request is a 128-B one, which
confirms the factor 2, and  hbm_bytes = read_bytes_by_size (or 2 x FETCH_SIZE) + WRITE_SIZE.
(Earlier summaries of this round calibrated FETCH_SIZE on fold_kernel's nominal bytes and got 1.67: that
kernel itself over-fetches by 20 % because its 1000-byte cell blocks straddle 128-byte lines.)
Infinity-Cache hits are included in these counters (the guide: "counted, not excluded")."""
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
rq = {c: per_kernel("rdreq", f"TCC_EA0_RDREQ_{c}sum") for c in ("", "32B_", "64B_", "128B_")}
res = {"units": "bytes per launch", "kernels": {}}
cal = None
for k in fetch:
    if "fold_kernel" in k:
        n_elem = fetch[k][2]  # grid size = threads >= n_cells*nd, rounded to 256
        known_read = 2 * 8 * n_elem
        cal = known_read / (fetch[k][0] * 1024)
        res["fold_kernel_check"] = {"kernel": k, "nominal_read_bytes": known_read, "FETCH_SIZE_KiB": fetch[k][0],
                                    "nominal_over_fetch_size": cal, "nominal_write_bytes": 8 * n_elem,
                                    "WRITE_SIZE_KiB": write.get(k, (0,))[0]}
for k in fetch:
    rd = fetch[k][0] * 1024
    wr = write.get(k, (0, 0, 0))[0] * 1024
    res["kernels"][k] = {"launches": fetch[k][1], "FETCH_SIZE_bytes_raw": rd, "WRITE_SIZE_bytes": wr,
                         "read_bytes_2x_fetch": 2 * rd,
                         "hbm_bytes": 2 * rd + wr,
                         "l2_hit_rate": (hit[k][0] / (hit[k][0] + miss[k][0])) if k in hit and hit[k][0] + miss[k][0] > 0 else None}
for k in res["kernels"]:
    if k in rq[""]:
        n, n32, n64, n128 = (rq[c].get(k, (0.0,))[0] for c in ("", "32B_", "64B_", "128B_"))
        # requests that are none of the three sized classes are counted at 64 B (FETCH_SIZE's own weight)
        by_size = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(n - n32 - n64 - n128, 0.0)
        res["kernels"][k]["rdreq"] = {"total": n, "32B": n32, "64B": n64, "128B": n128, "read_bytes_by_size": by_size}
        res["kernels"][k]["hbm_bytes"] = by_size + res["kernels"][k]["WRITE_SIZE_bytes"]
json.dump(res, open(out + "_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
