"""GPU parity of the general-Jacobian path (the reference's default geometry path,
fee_gpu.cuh:235-241,275-281): apply_batches_g through the C-ABI against the oracle on descriptions with a full
inverse Jacobian per quadrature point."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from util import deform, deformed_oracle_desc, desc_from_oracle, oracle_desc_from_mesh
from test_gpu import gpu_vmult, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nt,tol", [(mf.F64, 1e-12), (mf.F32, 2e-5)])
@pytest.mark.parametrize("p,n", [(1, 5), (2, 4), (3, 3), (4, 2), (4, 5), (5, 2), (6, 2), (2, 9)])
def test_general_jacobian_vmult_matches_oracle(p, n, nt, tol):
    od = deformed_oracle_desc(p, n, seed=p * 10 + n)
    desc, keep = desc_from_oracle(od, number_type=nt)
    assert not (desc.flags & mf.UNIFORM_J0)
    op = mf.Operator(desc, keep)
    assert op.kernel_name() == "apply_batches_g"
    rng = np.random.default_rng(7)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    if nt == mf.F32:  # the oracle computes in double on the float-rounded inputs
        od = o.Desc(3, p, od.n_dofs, od.loc2glob, od.JxW.astype(np.float32), od.inv_jac.astype(np.float32),
                    od.coefficient.astype(np.float32), od.constrained, None, np.float64,
                    od.shape_values.astype(np.float32), od.shape_gradients.astype(np.float32))
        x, y0 = x.astype(np.float32).astype(np.float64), y0.astype(np.float32).astype(np.float64)
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, x)) <= tol
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0, x)) <= tol


@pytest.mark.parametrize("p,n,cells", [(2, 8, 64), (4, 6, 27), (3, 6, 40)])
def test_general_jacobian_multi_cell_batches(p, n, cells):
    """batches of several chunks of cells (the default on large meshes; small meshes default to one cell per batch)"""
    od = deformed_oracle_desc(p, n, seed=p)
    desc, keep = desc_from_oracle(od, max_cells_per_batch=cells)
    op = mf.Operator(desc, keep)
    st = op.plan_stats()
    assert op.kernel_name() == "apply_batches_g" and st["n_batches"] >= 2 and st["max_batch_cells"] > 256 // (p + 1) ** 2
    rng = np.random.default_rng(11)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12


def test_general_jacobian_against_assembled_matrix_and_symmetry():
    od = deformed_oracle_desc(3, 3, eps=0.2, seed=3)
    desc, keep = desc_from_oracle(od)
    op = mf.Operator(desc, keep)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    A = o.assemble(od)
    ax, ay = gpu_vmult(op, x), gpu_vmult(op, y)
    assert rel(ax, A @ x) <= 1e-12
    assert abs(y @ ax - x @ ay) <= 1e-11 * abs(y @ ax)     # the operator is symmetric


def test_general_jacobian_reduces_to_the_cartesian_path():
    """inv_jac = (1/h) I at every point: both paths must give the same operator."""
    od = o.uniform_mesh_desc(3, 4, 4)
    nc, nd = od.n_cells, od.nd
    J = np.broadcast_to(np.eye(3) * od.inv_jac.reshape(nc, 1, 1, 1), (nc, nd, 3, 3)).copy()
    odg = o.Desc(3, 4, od.n_dofs, od.loc2glob, od.JxW, J, od.coefficient, od.constrained, None, np.float64,
                 od.shape_values, od.shape_gradients)
    x = np.random.default_rng(2).standard_normal(od.n_dofs)
    d1, k1 = desc_from_oracle(od)
    d2, k2 = desc_from_oracle(odg)
    y1 = gpu_vmult(mf.Operator(d1, k1), x)
    y2 = gpu_vmult(mf.Operator(d2, k2), x)
    assert rel(y2, y1) <= 1e-13


def test_general_jacobian_unsupported_combinations_are_loud():
    od = deformed_oracle_desc(2, 3)
    desc, keep = desc_from_oracle(od, colored=True)
    with pytest.raises(mf.MfgpuError):
        mf.Operator(desc, keep)


@pytest.mark.parametrize("p,n", [(1, 4), (2, 3), (4, 3), (5, 2)])
def test_general_jacobian_inverse_diagonal_and_pcg(p, n):
    """N1 on the N3 path: inverse diagonal against the oracle and the assembled matrix, then the Jacobi-PCG of
    test_gpu_aux on the deformed geometry against a sparse direct solve."""
    import scipy.sparse.linalg as spla

    od = deformed_oracle_desc(p, n, seed=5)
    desc, keep = desc_from_oracle(od)
    op = mf.Operator(desc, keep)
    N = od.n_dofs
    dinv = mf.DeviceVector(N)
    op.compute_inverse_diagonal(dinv)
    mf.synchronize()
    np.testing.assert_allclose(dinv.to_host(), o.compute_inverse_diagonal(od), rtol=1e-12)
    A = o.assemble(od).tocsc()
    np.testing.assert_allclose(1.0 / dinv.to_host(), A.diagonal(), rtol=1e-11)
    b_host = np.random.default_rng(1).standard_normal(N)
    b_host[od.constrained] = 0.0
    x_ref = spla.spsolve(A, b_host)
    b, x, r, z, pv, q = (mf.DeviceVector(N) for _ in range(6))
    b.from_host(b_host)
    r.equ(1.0, b)
    z.equ(1.0, r)
    z.scale(dinv)
    pv.equ(1.0, z)
    rz, r0 = r.dot(z), r.l2_norm()
    for its in range(1, 3000):
        op.vmult(q, pv)
        alpha = rz / pv.dot(q)
        x.add(alpha, pv)
        r.add(-alpha, q)
        if r.l2_norm() <= 1e-12 * r0:
            break
        z.equ(1.0, r)
        z.scale(dinv)
        rz_new = r.dot(z)
        pv.sadd(rz_new / rz, 1.0, z)
        rz = rz_new
    assert its < 2999
    assert np.linalg.norm(x.to_host() - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


@pytest.mark.parametrize("p,nref", [(1, 4), (2, 4), (4, 4), (3, 5)])
def test_general_jacobian_with_hanging_nodes(p, nref):
    """N3 on an adaptive mesh: the bmop ADAPTIVE_GRID mesh (hanging-node masks, substituted dof indices) with a
    full inverse Jacobian per quadrature point; vmult, vmult_add and the inverse diagonal against the oracle."""
    mesh = mf.Mesh.adaptive(3, p, nref)
    od = deform(oracle_desc_from_mesh(mesh, dtype=np.float64), seed=nref)
    assert od.constraint_mask is not None and not od.uniform_j0
    desc, keep = desc_from_oracle(od)
    assert desc.flags & mf.HANGING_NODES and not (desc.flags & mf.UNIFORM_J0)
    op = mf.Operator(desc, keep)
    assert op.kernel_name() == "apply_batches_g"
    rng = np.random.default_rng(11)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12
    dinv = mf.DeviceVector(od.n_dofs)
    op.compute_inverse_diagonal(dinv)
    mf.synchronize()
    np.testing.assert_allclose(dinv.to_host(), o.compute_inverse_diagonal(od), rtol=1e-12)
