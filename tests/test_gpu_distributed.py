"""Row (e) with the real local operator: two processes, each with its own z-slab, the HIP operator through
the C-ABI on device tensors and pymfgpu.parallel's exchange on those device tensors.  There is one GPU on the
test box, so both ranks use cuda:0 and the collective runs over gloo (which stages device tensors through the
host); gloo has no device-tensor send/recv, so the modes exercised are "pair" and "allreduce" -- the default
"p2p" mode shares everything but the transport call with them and is covered on CPU tensors in
tests/test_distributed.py.  RCCL itself needs two GPUs and is exercised only by the driver's scaling runs."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, p, n, q):
    for pth in (ROOT, os.path.join(ROOT, "dealii-cuda_amd"), os.path.join(ROOT, "tests")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import torch
    import torch.distributed as dist

    import pymfgpu as mf
    from pymfgpu.parallel import DistributedLaplace, slab_ranges

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    zb, ze = slab_ranges(n, world)[rank]
    mesh = mf.Mesh.uniform(3, p, n, slab=(zb, ze))
    gop = mf.Operator(mesh.desc, mesh)
    stream = torch.cuda.current_stream().cuda_stream

    def local_vmult(dst, src):
        gop.vmult(dst, src, stream)

    op = DistributedLaplace(mesh, rank, world, dev, torch.float64, local_vmult, mode)
    xyz = mesh.dof_coords()
    src = torch.from_numpy(np.sin(3 * xyz[:, 0]) + xyz[:, 1] ** 2 - np.cos(2 * xyz[:, -1]) * xyz[:, 0]).to(dev)
    dst = torch.full_like(src, 7.0)
    op.vmult(dst, src)
    dst2 = torch.zeros_like(src)
    op.vmult(dst2, dst)
    torch.cuda.synchronize()
    q.put((rank, xyz, dst.cpu().numpy().copy(), dst2.cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["pair", "allreduce"])
def test_two_slabs_gpu_operator_and_device_tensor_exchange(mode):
    import torch.multiprocessing as mp

    for pth in (ROOT, os.path.join(ROOT, "dealii-cuda_amd"), os.path.join(ROOT, "tests")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    import pymfgpu as mf
    from oracle import mf_oracle as o
    from util import oracle_desc_from_mesh

    world, p, n = 2, 4, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + (1 if mode == "pair" else 2)
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, p, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    full = mf.Mesh.uniform(3, p, n)
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
