"""The library's RCCL transport path (mfgpu_dist.hip: ncclCommInitRank, grouped ncclSend / ncclRecv with both
z-neighbours on the side stream, the events around them, the one-call mfgpu_vmult_dist) with TWO / THREE PROCESSES ON
ONE GPU.  RCCL refuses several ranks of a communicator on one device, so the seven RCCL entry points are replaced by
a test double (tests/fake_rccl/fake_rccl.cpp, LD_PRELOAD) that moves the planes through shared memory; everything
else -- the slab operators, the interface-first schedule, pack, streams, events, masked add -- is the product's code,
in separate processes as in a multi-GPU run.  Three chained distributed applies against the single-domain oracle
(relative l2 <= 1e-12, growing by ||A|| ~ 100 per apply)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from util import oracle_desc_from_mesh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "fake_rccl")


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call(["hipcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "fake_rccl.cpp"), "-lrt"],
                          stderr=subprocess.DEVNULL)
    return so


@pytest.mark.parametrize("p,n,world", [(4, 12, 2), (4, 9, 3), (2, 12, 3)])
def test_rccl_transport_path_between_processes(p, n, world, fake_rccl, tmp_path):
    idfile = str(tmp_path / "uid.bin")
    env = dict(os.environ, LD_PRELOAD=fake_rccl)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "worker.py"), str(r), str(world), str(p), str(n), idfile,
                               str(tmp_path / f"out{r}.npz")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [pr.communicate(timeout=300)[0] for pr in procs]
    for pr, out in zip(procs, outs):
        assert pr.returncode == 0, out[-3000:]
    full = mf.Mesh.uniform(3, p, n)
    od = oracle_desc_from_mesh(full)
    fx = full.dof_coords()
    key = {tuple(np.round(c, 9)): i for i, c in enumerate(fx)}
    ref = [np.sin(3 * fx[:, 0]) + fx[:, 1] ** 2 - np.cos(2 * fx[:, 2]) * fx[:, 0]]
    for _ in range(3):
        ref.append(o.vmult(od, ref[-1]))
    first = []
    for r in range(world):
        d = np.load(str(tmp_path / f"out{r}.npz"))
        gi = np.array([key[tuple(np.round(c, 9))] for c in d["xyz"]])
        for it in range(3):
            err = np.linalg.norm(d[f"y{it}"] - ref[it + 1][gi]) / np.linalg.norm(ref[it + 1])
            assert err <= 1e-12 * 100 ** it, (r, it, err)
        first.append(bool(d["schedule"][0]))
    if (p, n, world) == (4, 12, 2):
        assert all(first)  # thick slabs: the interface-first schedule ran over this transport
