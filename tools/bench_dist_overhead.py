#!/usr/bin/env python3
"""What the distributed schedule itself costs, without a network: three z-slabs of the C2 mesh on ONE GPU, connected by the
in-process transport (device copies instead of RCCL).  Compares, per apply of all slabs,
  plain : K x {vmult on all slabs}
  begin : K x {mfgpu_vmult_dist_begin on all slabs} -- interface batches first (one launch with a hole), priority pass 2
          and pack on the side stream, interior batches, rest of pass 2: what a rank does while the planes travel
  full  : K x {begin on all; end on all} -- adds the end phase; its device copies (2 per interface, each behind an
          event wait) are the stand-in for RCCL and do not exist in a multi-process run, the masked add (one launch) does
usage: tools/bench_dist_overhead.py [cells] [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import pymfgpu as mf  # noqa: E402
from pymfgpu.parallel import slab_ranges  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
world = 3
slabs = []
for r, (zb, ze) in enumerate(slab_ranges(n, world)):
    mesh = mf.Mesh.uniform(3, 4, n, slab=(zb, ze))
    op = mf.Operator(mesh.desc, mesh)
    d = mf.Dist(mesh, r, world)
    d.attach(op)
    a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
    a.fill(0.1)
    slabs.append(dict(mesh=mesh, op=op, d=d, a=a, b=b))
for lo, up in zip(slabs, slabs[1:]):
    lo["d"].connect_local(up["d"])
print("schedules (interface_first, r1_end, r2_begin, n_batches):", [s["d"].schedule() for s in slabs])


def run(mode):
    for s in slabs:
        s["a"].fill(0.1)
    mf.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        if mode == "plain":
            for s in slabs:
                s["op"].vmult(s["b"], s["a"])
            continue
        for s in slabs:
            s["d"].vmult_begin(s["op"], s["b"], s["a"])
        if mode == "full":
            for s in slabs:
                s["d"].vmult_end(s["op"], s["b"])
    mf.synchronize()
    if mode == "begin":  # leave no exchange open
        for s in slabs:
            s["d"].vmult_end(s["op"], s["b"])
        mf.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for m in ("full", "begin", "plain"):
    run(m)
for rep in range(3):
    tf, tb, tp = run("full"), run("begin"), run("plain")
    print(f"per apply of all {world} slabs: plain {tp:.4f} ms, begin {tb:.4f}, full {tf:.4f}; per slab: begin phase "
          f"+{1e3 * (tb - tp) / world:.1f} us, end phase (in-process copies + add) +{1e3 * (tf - tb) / world:.1f} us")
