#!/usr/bin/env python3
"""Eager launches vs one captured hipGraph for the bmop loop (K x {swap; vmult}) on launch-bound and on large problems.
usage: tools/bench_graph.py   -> one JSON line per configuration"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import pymfgpu as mf  # noqa: E402

K = 100
for name, dim, p, n in (("C1 2D p=2 32^2 cells", 2, 2, 32), ("3D p=4 8^3 cells", 3, 4, 8), ("3D p=4 24^3 cells", 3, 4, 24),
                        ("C2 3D p=4 54^3 cells", 3, 4, 54)):
    mesh = mf.Mesh.uniform(dim, p, n)
    op = mf.Operator(mesh.desc, mesh)
    N = mesh.n_dofs
    a = torch.full((N,), 0.1, device="cuda", dtype=torch.float64)
    b = torch.zeros(N, device="cuda", dtype=torch.float64)

    def loop(stream, k):
        x, y = a, b
        for _ in range(k):
            op.vmult(y, x, stream)
            x, y = y, x
            y.mul_(1e-3)  # keep values finite (a cheap kernel, also captured)

    st = torch.cuda.current_stream().cuda_stream
    loop(st, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(st, K)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loop(torch.cuda.current_stream().cuda_stream, K)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / K
    print(json.dumps({"config": name, "n_dofs": N, "eager_us_per_vmult": 1e6 * eager, "graph_us_per_vmult": 1e6 * graph}))
