#!/usr/bin/env python3
"""Measurement of the SURVEY.md 8(f) rows N1 / N2 on the headline mesh (p=4 3D, n=54, 10.2 M dofs, double):
achieved GB/s of the vector kernels against their algorithmic bytes, the inverse-diagonal setup time and one
Jacobi-PCG iteration.  HIP-event-free on purpose (no torch): wall clock around K back-to-back calls."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import numpy as np  # noqa: E402
import pymfgpu as mf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
mesh = mf.Mesh.uniform(3, 4, n)
op = mf.Operator(mesh.desc, mesh)
N = mesh.n_dofs
rng = np.random.default_rng(0)
v, w, x, d = (mf.DeviceVector(N) for _ in range(4))
for t in (v, w, x):
    t.from_host(rng.standard_normal(N))


def timed(fn, k=50):
    fn()
    mf.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    mf.synchronize()
    return (time.perf_counter() - t0) / k


res = {}
for name, fn, nbytes in (("sadd", lambda: v.sadd(0.999, 1e-3, w), 24 * N), ("equ", lambda: v.equ(1.0, w), 16 * N),
                         ("scale", lambda: v.scale(w), 24 * N), ("dot", lambda: v.dot(w), 16 * N),
                         ("l2_norm", lambda: v.l2_norm(), 8 * N),
                         ("add_and_dot", lambda: v.add_and_dot(1e-6, x, w), 32 * N)):
    t = timed(fn)
    res[name] = {"us": 1e6 * t, "GB/s": nbytes / t / 1e9, "frac_of_8TBs": nbytes / t / 8e12}
res["compute_inverse_diagonal"] = {"us": 1e6 * timed(lambda: op.compute_inverse_diagonal(d), 5)}
res["vmult"] = {"us": 1e6 * timed(lambda: op.vmult(v, w), 50)}


def pcg_iteration():
    op.vmult(x, w)
    a = 1e-9 / (1.0 + abs(w.dot(x)))
    v.add(a, w)
    d.add(-a, x)
    d.l2_norm()
    x.equ(1.0, d)
    x.scale(d)
    w.sadd(1e-9, 1.0, x)


op.compute_inverse_diagonal(d)
res["pcg_iteration"] = {"us": 1e6 * timed(pcg_iteration, 20)}
print(json.dumps({"workload": f"p4_3d_n{n}_f64", "n_dofs": N, "results": res}))
