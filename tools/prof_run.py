#!/usr/bin/env python3
"""Profiling target: N vmult applies of the headline workload, no torch (run under rocprofv3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import pymfgpu as mf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p = int(sys.argv[4]) if len(sys.argv) > 4 else 4
nt = mf.F32 if len(sys.argv) > 5 and sys.argv[5] == "f32" else mf.F64
mesh = mf.Mesh.uniform(3, p, n, number_type=nt)
mesh.desc.max_cells_per_batch = bc
op = mf.Operator(mesh.desc, mesh)
dst, src = mf.DeviceVector(mesh.n_dofs, nt), mf.DeviceVector(mesh.n_dofs, nt)
dst.fill(0.1)
for i in range(steps):
    dst.swap(src)
    op.vmult(dst, src)
    if i % 8 == 7:
        dst.fill(0.1)  # keep values finite
mf.synchronize()
print("done", mesh.n_dofs, op.plan_stats())
