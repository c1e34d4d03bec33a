"""SURVEY.md 8(f) N1 on the CPU: the oracle's inverse diagonal against the assembled matrix
(test_laplace_op.cu:50-120 procedure) -- pins oracle.compute_inverse_diagonal independently of the product."""
import numpy as np
import pytest

from oracle import mf_oracle as o


@pytest.mark.parametrize("dim,p,n", [(2, 1, 5), (2, 2, 4), (2, 4, 3), (3, 1, 3), (3, 2, 3), (3, 4, 2)])
def test_inverse_diagonal_equals_assembled_diagonal(dim, p, n):
    od = o.uniform_mesh_desc(dim, p, n)
    inv = o.compute_inverse_diagonal(od)
    A = o.assemble(od)  # constrained rows/columns are identity: diagonal 1 there, as set_constrained_values(.,1)
    np.testing.assert_allclose(1.0 / inv, A.diagonal(), rtol=1e-12)
    assert (inv[od.constrained] == 1.0).all()


def test_local_diagonal_is_diag_of_the_cell_kernel():
    """DiagonalLocalOperator's own procedure (laplace_operator_gpu.h:374-392): apply the cell kernel to
    every local unit vector and keep entry i -- against the closed form used by oracle.local_diagonal."""
    od = o.uniform_mesh_desc(3, 2, 2)
    loc = o.local_diagonal(od)
    nd = od.nd
    brute = np.zeros_like(loc)
    for i in range(nd):
        e = np.zeros((od.n_cells, nd))
        e[:, i] = 1.0
        brute[:, i] = o.cell_apply(od, e)[:, i]
    np.testing.assert_allclose(loc, brute, rtol=1e-13)


def test_inverse_diagonal_with_hanging_node_masks_follows_the_reference_quirk():
    """With hanging nodes the reference distributes the local diagonal VECTOR through the transposed
    constraint (C^T diag(K)), which is NOT diag(C^T K C); the oracle restates the reference.  Parity
    unpinned beyond that: the reference holds no fixture for compute_diagonal."""
    od = o.uniform_mesh_desc(2, 2, 4)
    cm = np.zeros(od.n_cells, dtype=np.uint32)
    cm[5] = (1 << 4) | (1 << 0)   # a face mask on one cell (hanging_nodes.cuh:38-50)
    od.constraint_mask = cm
    inv = o.compute_inverse_diagonal(od)
    true_diag = o.assemble(od).diagonal()
    g = od.loc2glob[5]
    assert np.isfinite(inv).all()
    assert np.abs(1.0 / inv[g] - true_diag[g]).max() > 1e-8      # the quirk is visible on that cell
    other = np.setdiff1d(np.arange(od.n_dofs), g)
    np.testing.assert_allclose(1.0 / inv[other], true_diag[other], rtol=1e-12)


@pytest.mark.parametrize("p,n", [(1, 3), (2, 2), (3, 2)])
def test_general_jacobian_oracle_against_assembled_matrix(p, n):
    """SURVEY.md 8f N3 on the CPU: the oracle's general-geometry branch (fee_gpu.cuh:235-241,275-281 restated
    in cell_apply) against the independently assembled matrix (physical gradients, test_laplace_op.cu:50-120)."""
    from util import deformed_oracle_desc

    od = deformed_oracle_desc(p, n)
    assert not od.uniform_j0
    x = np.random.default_rng(1).standard_normal(od.n_dofs)
    y = o.vmult(od, x)
    ref = o.assemble(od) @ x
    assert np.linalg.norm(y - ref) <= 1e-13 * np.linalg.norm(ref)


def test_inverse_diagonal_general_geometry_equals_assembled_diagonal():
    from util import deformed_oracle_desc

    od = deformed_oracle_desc(2, 2)
    np.testing.assert_allclose(1.0 / o.compute_inverse_diagonal(od), o.assemble(od).diagonal(), rtol=1e-12)
