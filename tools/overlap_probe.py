#!/usr/bin/env python3
"""Diagnostic (diag build): does the CU overlap a memory-only and a cells-only instance of the kernel
when they run concurrently on two streams, one workgroup per CU each?  MFGPU_DBG is read per call."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MFGPU_LIB"] = os.path.join(ROOT, "dealii-cuda_amd", "lib", "libmfgpu_diag.so")
os.environ["MFGPU_GRID"] = "256"
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import torch  # noqa: E402

import pymfgpu as mf  # noqa: E402

mesh = mf.Mesh.uniform(3, 4, 54)
ops = [mf.Operator(mesh.desc, mesh) for _ in range(2)]
N = mesh.n_dofs
vec = [(torch.full((N,), 0.1, device="cuda", dtype=torch.float64), torch.zeros(N, device="cuda", dtype=torch.float64))
       for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(dbgs, K=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for i, d in enumerate(dbgs):
            if d is None:
                continue
            os.environ["MFGPU_DBG"] = str(d)
            ops[i].vmult(vec[i][1], vec[i][0], streams[i].cuda_stream)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / K


for name, dbgs in [("full alone (grid 256)", (0, None)), ("memory-only alone", (1, None)), ("cells-only alone", (14, None)),
                   ("memory-only || cells-only", (1, 14)), ("memory-only || memory-only", (1, 1)),
                   ("cells-only || cells-only", (14, 14)), ("full || full", (0, 0))]:
    run(dbgs, 3)
    print(f"{name:32s} {run(dbgs):8.1f} us per (pair of) vmult")
