"""GPU tests of the segmented cell loop (mfgpu_desc.cell_loop_segments): the cell loop as several launches over
consecutive batch ranges with pass 2 of the finished ranges on the handle's side stream.  The summation order of
every dof is the same whatever the segmentation (ascending batch order), so results must be BIT-IDENTICAL to the
one-launch form, and within the usual tolerance of the oracle (double 1e-12, float 1e-5 relative l2)."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from test_gpu import TOL, gpu_vmult, rel
from util import deformed_oracle_desc, desc_from_oracle, oracle_desc_from_mesh

pytestmark = pytest.mark.gpu


def _same(a, b, exact, nt):
    # apply_planes3 (one wave per batch, in-order LDS adds) is bit-reproducible; the pencil kernels sum the cells of a
    # batch with LDS atomics from several waves, whose order varies from launch to launch in the last bit
    if exact:
        np.testing.assert_array_equal(a, b)
    else:
        assert rel(a, b) <= (1e-14 if nt == mf.F64 else 1e-6)


def _check(make_op, od, n_dofs, nt=mf.F64, segs=(2, 3, 7), exact=False):
    rng = np.random.default_rng(n_dofs)
    x, y0 = rng.standard_normal(n_dofs), rng.standard_normal(n_dofs)
    xt = x.astype(mf.np_dtype(nt)).astype(np.float64)
    y0t = y0.astype(mf.np_dtype(nt)).astype(np.float64)
    op1 = make_op(1)
    ref, ref_add = gpu_vmult(op1, x, nt), gpu_vmult(op1, x, nt, y0=y0)
    assert rel(ref, o.vmult(od, xt)) <= TOL[nt]
    assert rel(ref_add, o.vmult_add(od, y0t, xt)) <= TOL[nt]
    for s in (0,) + tuple(segs):
        op = make_op(s)
        _same(gpu_vmult(op, x, nt), ref, exact, nt)
        _same(gpu_vmult(op, x, nt, y0=y0), ref_add, exact, nt)
        # back-to-back applies on one handle: the next cell loop must not overtake the side stream's pass 2
        a, b = mf.DeviceVector(n_dofs, nt), mf.DeviceVector(n_dofs, nt)
        a.from_host(x)
        for _ in range(4):
            op.vmult(b, a)
            op.vmult(a, b)
        mf.synchronize()
        got = a.to_host()
        a.from_host(x)
        for _ in range(4):
            op1.vmult(b, a)
            op1.vmult(a, b)
        mf.synchronize()
        _same(got, a.to_host(), exact, nt)


@pytest.mark.parametrize("dim,p,n,kern,nt", [(3, 4, 6, 0, mf.F64), (3, 4, 7, 0, mf.F32), (3, 2, 9, 0, mf.F64),
                                              (3, 3, 7, mf.KERNEL_PLANES, mf.F64), (3, 5, 4, 0, mf.F64),
                                              (2, 2, 16, 0, mf.F64), (2, 4, 9, 0, mf.F32)])
def test_segments_uniform(dim, p, n, kern, nt):
    mesh = mf.Mesh.uniform(dim, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)

    def make_op(s):
        mesh.desc.kernel = kern
        mesh.desc.cell_loop_segments = s
        mesh.desc.max_cells_per_batch = 0 if (dim == 3 and p == 4) else 8  # several batches on these small meshes
        return mf.Operator(mesh.desc, mesh)

    assert make_op(1).plan_stats()["n_batches"] >= 8
    _check(make_op, od, mesh.n_dofs, nt, exact=make_op(1).kernel_name() == "apply_planes3")


@pytest.mark.parametrize("p,nref", [(4, 4), (2, 4), (4, 5)])
def test_segments_adaptive(p, nref):
    """hanging nodes: plane batches + pencil batches; the default cut is the family boundary"""
    mesh = mf.Mesh.adaptive(3, p, nref)
    od = oracle_desc_from_mesh(mesh)

    def make_op(s):
        mesh.desc.cell_loop_segments = s
        mesh.desc.max_cells_per_batch = 8
        return mf.Operator(mesh.desc, mesh)

    _check(make_op, od, mesh.n_dofs, segs=(2, 5))


def test_segments_general_geometry():
    od = deformed_oracle_desc(3, 5, eps=0.1, seed=3)

    def make_op(s):
        desc, keep = desc_from_oracle(od, max_cells_per_batch=8, cell_loop_segments=s)
        return mf.Operator(desc, keep)

    _check(make_op, od, od.n_dofs, segs=(2, 4))
