#!/usr/bin/env python3
"""Timeline of the last few mfgpu kernel dispatches of a rocprofv3 --kernel-trace run: start / end in us relative to
the first listed dispatch, queue id.  usage: tools/trace_timeline.py DIR [N]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mfgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mfgpu::", "").replace("(anonymous namespace)::", "")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name[:40]:40s} q={r.get('Queue_Id', '?'):>3s} grid={r.get('Grid_Size', '?'):>8s} start={s / 1e3:9.1f} end={e / 1e3:9.1f} dur={(e - s) / 1e3:8.1f}")
