"""One rank of tests/test_gpu_dist_rccl_path.py: a z-slab operator + mfgpu_dist on the RCCL transport path (the RCCL
entry points are tests/fake_rccl/fake_rccl.cpp, preloaded), three chained distributed applies."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import pymfgpu as mf  # noqa: E402
from pymfgpu.parallel import slab_ranges  # noqa: E402

rank, world, p, n = (int(a) for a in sys.argv[1:5])
idfile, outfile = sys.argv[5], sys.argv[6]
if rank == 0:
    uid = mf.dist_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
zb, ze = slab_ranges(n, world)[rank]
mesh = mf.Mesh.uniform(3, p, n, slab=(zb, ze))
op = mf.Operator(mesh.desc, mesh)
dist = mf.Dist(mesh, rank, world, unique_id=uid)  # ncclCommInitRank: returns when every rank has arrived
dist.attach(op)
xyz = mesh.dof_coords()
x = np.sin(3 * xyz[:, 0]) + xyz[:, 1] ** 2 - np.cos(2 * xyz[:, 2]) * xyz[:, 0]
a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
a.from_host(x)
res = []
for it in range(3):
    dist.vmult(op, b, a)  # the one-call form: cell loop, exchange over the (fake) RCCL calls, masked add
    mf.synchronize()
    res.append(b.to_host().copy())
    a, b = b, a
np.savez(outfile, xyz=xyz, y0=res[0], y1=res[1], y2=res[2], schedule=np.array(dist.schedule(), dtype=np.int64))
