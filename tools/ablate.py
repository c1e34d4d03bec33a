#!/usr/bin/env python3
"""Ablation timing with the diagnostic build: MFGPU_DBG bits 1 cells, 2 gather, 4 scatter, 8 prefetch."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ["MFGPU_LIB"] = os.path.join(ROOT, "dealii-cuda_amd", "lib", "libmfgpu_diag.so")
    sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
    import pymfgpu as mf
    n, bc = int(sys.argv[2]), int(sys.argv[3])
    mesh = mf.Mesh.uniform(3, 4, n)
    mesh.desc.max_cells_per_batch = bc
    op = mf.Operator(mesh.desc, mesh)
    dst, src = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
    src.fill(0.1)
    for _ in range(5):
        op.vmult(dst, src)
    mf.synchronize()
    t0 = time.perf_counter()
    K = 30
    for _ in range(K):
        op.vmult(dst, src)
    t1 = time.perf_counter()
    mf.synchronize()
    print(f"dbg={os.environ.get('MFGPU_DBG','0'):>2s} bc={bc:2d}: {1e6*(time.perf_counter()-t0)/K:8.1f} us/vmult  (host enqueue {1e6*(t1-t0)/K:8.1f} us/vmult)")
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    for bc in (int(os.environ.get("ABL_BC", "27")),):
        for dbg in [int(x) for x in os.environ.get("ABL_DBG", "0,2,8,4,10,14,30").split(",")]:
            env = dict(os.environ, MFGPU_DBG=str(dbg))
            subprocess.run([sys.executable, __file__, "child", str(n), str(bc)], env=env)
