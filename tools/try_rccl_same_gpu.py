#!/usr/bin/env python3
"""Probe: does RCCL accept two ranks of one communicator on ONE device (the single-GPU development box)?
If yes, the RCCL transport of mfgpu_dist can be exercised there; if not ("Duplicate GPU detected"), only on a real
multi-GPU node.  Prints the outcome; never part of the test suite."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))


def worker(rank, world, idq):
    import numpy as np
    import pymfgpu as mf
    n, p = 6, 2
    from pymfgpu.parallel import slab_ranges
    zb, ze = slab_ranges(n, world)[rank]
    mesh = mf.Mesh.uniform(3, p, n, slab=(zb, ze))
    if rank == 0:
        uid = mf.dist_unique_id()
        for _ in range(world - 1):
            idq.put(uid)
    else:
        uid = idq.get(timeout=30)
    try:
        d = mf.Dist(mesh, rank, world, unique_id=uid)
    except Exception as e:  # noqa: BLE001
        print(f"rank {rank}: Dist failed: {e}", flush=True)
        return
    op = mf.Operator(mesh.desc, mesh)
    d.attach(op)
    a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
    a.fill(0.1)
    d.vmult(op, b, a)
    mf.synchronize()
    print(f"rank {rank}: distributed vmult over RCCL on one device OK, |y| = {np.linalg.norm(b.to_host()):.6e}", flush=True)


if __name__ == "__main__":
    mp.set_start_method("spawn")
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(r, 2, q)) for r in range(2)]
    for pr in ps:
        pr.start()
    for pr in ps:
        pr.join(60)
        if pr.is_alive():
            print("timeout: killing", pr.pid)
            pr.terminate()
