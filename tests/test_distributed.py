"""N>1 path on CPU: world_size-2/3 gloo processes, z-slab meshes from the product's host code,
exchange code of pymfgpu.parallel, ORACLE as the local operator (the HIP kernel needs a GPU)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, dim, p, n, q):
    for pth in (ROOT, os.path.join(ROOT, "dealii-cuda_amd"), os.path.join(ROOT, "tests")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import torch
    import torch.distributed as dist

    import pymfgpu as mf
    from oracle import mf_oracle as o
    from pymfgpu.parallel import DistributedLaplace, slab_ranges
    from util import oracle_desc_from_mesh

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    zb, ze = slab_ranges(n, world)[rank]
    mesh = mf.Mesh.uniform(dim, p, n, slab=(zb, ze))
    od = oracle_desc_from_mesh(mesh)

    def local_vmult(dst, src):
        dst.copy_(torch.from_numpy(o.vmult(od, src.numpy())))

    op = DistributedLaplace(mesh, rank, world, "cpu", torch.float64, local_vmult, mode)
    # global input: function of the dof coordinates so every rank builds consistent ghost values
    xyz = mesh.dof_coords()
    src = torch.from_numpy(np.sin(3 * xyz[:, 0]) + xyz[:, 1] ** 2 - np.cos(2 * xyz[:, -1]) * xyz[:, 0])
    dst = torch.zeros_like(src)
    op.vmult(dst, src)
    # second apply on the exchanged result: ghost values must already be consistent
    dst2 = torch.zeros_like(src)
    op.vmult(dst2, dst)
    q.put((rank, xyz, dst.numpy().copy(), dst2.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "p2p"), (3, "p2p"), (2, "pair"), (3, "pair"), (2, "allreduce")])
def test_slab_partition_matches_single_domain(world, mode):
    import torch.multiprocessing as mp

    import pymfgpu as mf
    from oracle import mf_oracle as o
    from util import oracle_desc_from_mesh

    dim, p, n = 3, 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, dim, p, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    full = mf.Mesh.uniform(dim, p, n)
    od = oracle_desc_from_mesh(full)
    xyz = full.dof_coords()
    src = np.sin(3 * xyz[:, 0]) + xyz[:, 1] ** 2 - np.cos(2 * xyz[:, -1]) * xyz[:, 0]
    ref1 = o.vmult(od, src)
    ref2 = o.vmult(od, ref1)
    key = {tuple(np.round(c, 9)): i for i, c in enumerate(xyz)}
    for rank, cxyz, d1, d2 in res:
        gi = np.array([key[tuple(np.round(c, 9))] for c in cxyz])
        assert np.linalg.norm(d1 - ref1[gi]) <= 1e-12 * np.linalg.norm(ref1)
        assert np.linalg.norm(d2 - ref2[gi]) <= 1e-11 * np.linalg.norm(ref2)
