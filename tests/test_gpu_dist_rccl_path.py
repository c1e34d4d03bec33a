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
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "fake_rccl.cpp"), "-lrt", "-ldl"])
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


def test_bench_multi_rank_path_on_one_gpu(fake_rccl):
    """bench.py --gpus 2 end to end on the one GPU of the test box: the parent starts the two ranks itself, the ranks
    build their slabs, rendezvous the communicator id, check one distributed apply against the dense all-reduce of the
    interface planes, time the steps, and rank 0 prints the JSON line.  (Every rank on device 0, torch process group
    over gloo, the library's RCCL calls on the test double: MFGPU_BENCH_TEST_ONE_GPU.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LD_PRELOAD=fake_rccl, MFGPU_BENCH_TEST_ONE_GPU="1")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                         "--cells", "12", "--no-cpu", "--ramp-steps", "20"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-3000:]
    d = json.loads(pr.stdout.strip().split("\n")[-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "slab2/cxx" and d["config"]["finite"]
    assert d["config"]["cells_per_dir"] == 18 and d["value"] > 0  # 12 * 2^(1/3) to a multiple of 6 (weak scaling)


@pytest.mark.parametrize("fault", ["error", "corrupt"])
def test_bench_falls_back_when_the_rccl_path_fails(fake_rccl, fault):
    """the first multi-GPU run is also the first run of the RCCL transport: if its grouped send / recv fails, or its result
    disagrees with the dense all-reduce of the interface planes, bench.py times the torch.distributed exchange of the
    test double instead and says so in the line (the library closes the RCCL group it opened before it reports the
    error, so the process's later collectives are not deferred)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LD_PRELOAD=fake_rccl, MFGPU_BENCH_TEST_ONE_GPU="1", FAKE_RCCL_FAULT=fault)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                         "--cells", "12", "--no-cpu", "--ramp-steps", "20"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-3000:]
    d = json.loads(pr.stdout.strip().split("\n")[-1])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "slab2/p2p" and d["config"]["finite"] and d["value"] > 0
    assert "falling back to --mode p2p" in pr.stderr
