"""GPU parity on the BALL domain (bmop -DBALL_GRID, poisson_common.h:65-70): unstructured hyper_ball mesh, MappingQ1
geometry, general-Jacobian operator path (apply_batches_g) through the C-ABI against the oracle on the same arrays.
Tolerance: relative l2 <= 1e-12 in double, 2e-5 in float."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from test_gpu import gpu_vmult, rel
from util import oracle_desc_from_mesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p,n_ref,nt,cells", [(1, 2, mf.F64, 0), (2, 1, mf.F64, 0), (2, 2, mf.F64, 16), (3, 1, mf.F64, 8),
                                               (4, 1, mf.F64, 0), (4, 2, mf.F64, 27), (4, 1, mf.F32, 8), (6, 0, mf.F64, 0)])
def test_ball_vmult_matches_oracle(p, n_ref, nt, cells):
    mesh = mf.Mesh.ball(3, p, n_ref, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    mesh.desc.max_cells_per_batch = cells
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == "apply_batches_g"
    rng = np.random.default_rng(p * 10 + n_ref)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    tol = 1e-12 if nt == mf.F64 else 2e-5
    xt = x.astype(mf.np_dtype(nt)).astype(np.float64)
    y0t = y0.astype(mf.np_dtype(nt)).astype(np.float64)
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= tol
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= tol
    # Dirichlet rows are identity rows
    c = mesh.arrays()["constrained_dofs"]
    np.testing.assert_array_equal(gpu_vmult(op, x, nt)[c], x.astype(mf.np_dtype(nt))[c])


def test_ball_inverse_diagonal_and_symmetry():
    mesh = mf.Mesh.ball(3, 3, 1)
    od = oracle_desc_from_mesh(mesh)
    op = mf.Operator(mesh.desc, mesh)
    d = mf.DeviceVector(mesh.n_dofs)
    op.compute_inverse_diagonal(d)
    mf.synchronize()
    assert rel(d.to_host(), o.compute_inverse_diagonal(od)) <= 1e-12
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    ax, ay = gpu_vmult(op, x), gpu_vmult(op, y)
    assert abs(y @ ax - x @ ay) <= 1e-11 * abs(y @ ax)


def test_ball_bmop_protocol():
    """bmop.cu:134-146 on the ball: dst = 0.1; 3 x {swap; vmult}"""
    mesh = mf.Mesh.ball(3, 4, 1)
    od = oracle_desc_from_mesh(mesh)
    op = mf.Operator(mesh.desc, mesh)
    a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
    b.fill(0.1)
    for _ in range(3):
        a, b = b, a
        op.vmult(b, a)
    mf.synchronize()
    ref = o.bmop_protocol(od, 3)
    assert rel(b.to_host(), ref) <= 1e-12 * 100 ** 2
