"""GPU parity of the general-Jacobian path in 2D (fee_gpu.cuh:235-241,275-281 is dimension-generic):
apply_batches_g2 through the C-ABI against the oracle -- deformed Cartesian meshes, the 2D BALL domain (bmop
-DBALL_GRID -DDIMENSION=2), hanging nodes, inverse diagonal.  Tolerance: relative l2 <= 1e-12 (double), 2e-5 (float)."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from test_gpu import gpu_vmult, rel
from util import deform, desc_from_oracle, oracle_desc_from_mesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nt,tol", [(mf.F64, 1e-12), (mf.F32, 2e-5)])
@pytest.mark.parametrize("p,n,cells", [(1, 9, 0), (2, 8, 0), (2, 32, 0), (3, 7, 9), (4, 6, 0), (4, 12, 30), (5, 5, 4), (6, 4, 0)])
def test_general2d_vmult_matches_oracle(p, n, cells, nt, tol):
    od = deform(o.uniform_mesh_desc(2, p, n), eps=0.15, seed=p * 10 + n)
    desc, keep = desc_from_oracle(od, number_type=nt, max_cells_per_batch=cells)
    assert not (desc.flags & mf.UNIFORM_J0)
    op = mf.Operator(desc, keep)
    assert op.kernel_name() == "apply_batches_g2"
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    if nt == mf.F32:  # the oracle computes in double on the float-rounded inputs
        od = o.Desc(2, p, od.n_dofs, od.loc2glob, od.JxW.astype(np.float32), od.inv_jac.astype(np.float32),
                    od.coefficient.astype(np.float32), od.constrained, None, np.float64,
                    od.shape_values.astype(np.float32), od.shape_gradients.astype(np.float32))
        x, y0 = x.astype(np.float32).astype(np.float64), y0.astype(np.float32).astype(np.float64)
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, x)) <= tol
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0, x)) <= tol


def test_general2d_reduces_to_the_cartesian_path_and_matches_the_assembled_matrix():
    od = o.uniform_mesh_desc(2, 4, 5)
    nc, nd = od.n_cells, od.nd
    J = np.broadcast_to(np.eye(2) * od.inv_jac.reshape(nc, 1, 1, 1), (nc, nd, 2, 2)).copy()
    odg = o.Desc(2, 4, od.n_dofs, od.loc2glob, od.JxW, J, od.coefficient, od.constrained, None, np.float64,
                 od.shape_values, od.shape_gradients)
    x = np.random.default_rng(2).standard_normal(od.n_dofs)
    d1, k1 = desc_from_oracle(od)
    d2, k2 = desc_from_oracle(odg)
    assert rel(gpu_vmult(mf.Operator(d2, k2), x), gpu_vmult(mf.Operator(d1, k1), x)) <= 1e-13
    odd = deform(od, eps=0.2, seed=1)
    d3, k3 = desc_from_oracle(odd)
    A = o.assemble(odd)
    assert rel(gpu_vmult(mf.Operator(d3, k3), x), A @ x) <= 1e-12


@pytest.mark.parametrize("p,n_ref", [(1, 3), (2, 2), (4, 2), (4, 4), (6, 1)])
def test_general2d_ball(p, n_ref):
    """bmop -DBALL_GRID -DDIMENSION=2: hyper_ball (5 coarse cells), unstructured, MappingQ1"""
    mesh = mf.Mesh.ball(2, p, n_ref)
    od = oracle_desc_from_mesh(mesh)
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == "apply_batches_g2"
    rng = np.random.default_rng(p + n_ref)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12
    dinv = mf.DeviceVector(mesh.n_dofs)
    op.compute_inverse_diagonal(dinv)
    mf.synchronize()
    np.testing.assert_allclose(dinv.to_host(), o.compute_inverse_diagonal(od), rtol=1e-12)


@pytest.mark.parametrize("p,nref", [(1, 4), (2, 5), (4, 4), (3, 6)])
def test_general2d_with_hanging_nodes(p, nref):
    mesh = mf.Mesh.adaptive(2, p, nref)
    od = deform(oracle_desc_from_mesh(mesh, dtype=np.float64), seed=nref)
    assert od.constraint_mask is not None and np.count_nonzero(od.constraint_mask) > 0
    desc, keep = desc_from_oracle(od, max_cells_per_batch=12)
    op = mf.Operator(desc, keep)
    assert op.kernel_name() == "apply_batches_g2"
    rng = np.random.default_rng(11)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12
    dinv = mf.DeviceVector(od.n_dofs)
    op.compute_inverse_diagonal(dinv)
    mf.synchronize()
    np.testing.assert_allclose(dinv.to_host(), o.compute_inverse_diagonal(od), rtol=1e-12)
