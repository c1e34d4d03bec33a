"""Host-side tests of the BALL mesh stand-in (mfgpu_mesh_create_ball; reference poisson_common.h:65-70,
bmop_common.h:108-120 with -DBALL_GRID): topology (dof counts from an independent vertex / line / quad count), the
geometry's convergence to the unit ball, the Dirichlet set, and the oracle's own three-way differential on this
unstructured, non-affine mesh (sum-factorised general-Jacobian path == assembled matrix, test_laplace_op.cu:50-120).
deal.II is not available here: the generator restates the published recipe, parity with deal.II's vertices is unpinned."""
import math

import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from util import oracle_desc_from_mesh


def _entity_counts(dim, n_ref):
    """vertices, lines, quads, cells of hyper_ball after n_ref global refinements, from the recurrences of regular
    refinement (a line gives 2 lines + 1 vertex; a quad 4 quads + 4 lines + 1 vertex; a hex 8 hexes + 12 quads + 6
    lines + 1 vertex)"""
    if dim == 2:
        v, l, q = 8, 12, 5
        for _ in range(n_ref):
            v, l, q = v + l + q, 2 * l + 4 * q, 4 * q
        return v, l, q, 0
    v, l, q, h = 16, 32, 24, 7
    for _ in range(n_ref):
        v, l, q, h = v + l + q + h, 2 * l + 4 * q + 6 * h, 4 * q + 12 * h, 8 * h
    return v, l, q, h


@pytest.mark.parametrize("dim,p,n_ref", [(2, 1, 0), (2, 2, 2), (2, 4, 3), (3, 1, 1), (3, 2, 2), (3, 4, 1), (3, 3, 2)])
def test_ball_topology_and_dirichlet_set(dim, p, n_ref):
    mesh = mf.Mesh.ball(dim, p, n_ref)
    a = mesh.arrays()
    v, l, q, h = _entity_counts(dim, n_ref)
    assert mesh.n_cells == (q if dim == 2 else h)
    expect = v + (p - 1) * l + (p - 1) ** 2 * q + (p - 1) ** 3 * h
    assert mesh.n_dofs == expect
    l2g = a["loc2glob"]
    assert l2g.max() == mesh.n_dofs - 1 and len(np.unique(l2g)) == mesh.n_dofs
    # every cell lists nd DIFFERENT dofs
    assert all(len(np.unique(row)) == mesh.nd for row in l2g)
    # conformity: a dof has ONE position in space whichever cell computes it (the generator stores the last cell's)
    xc = mesh.dof_coords()
    od = oracle_desc_from_mesh(mesh)
    nodes = o.shape_info(p)[4]
    # Dirichlet set == dofs on the unit sphere (the boundary vertices are ON the sphere; boundary faces are flat
    # between them, so other boundary dofs lie slightly inside: test through the cells' boundary faces instead)
    r = np.linalg.norm(xc, axis=1)
    con = np.zeros(mesh.n_dofs, bool)
    con[a["constrained_dofs"]] = True
    assert con.sum() > 0 and np.all(r[con] > 0.5) and np.all(r[con] <= 1 + 1e-12)
    # a dof strictly inside the ball's polyhedral approximation is never constrained: all dofs with |x| below the
    # smallest constrained radius are free, and every dof ON the sphere is constrained
    assert np.all(con[np.abs(r - 1.0) < 1e-12])
    assert not np.any(con[r < r[con].min() - 1e-12])
    assert od.n_dofs == mesh.n_dofs and nodes.size == p + 1


@pytest.mark.parametrize("dim", [2, 3])
def test_ball_geometry_converges(dim):
    exact = math.pi if dim == 2 else 4.0 * math.pi / 3.0
    err = []
    for n_ref in (1, 2, 3):
        mesh = mf.Mesh.ball(dim, 2, n_ref)  # (the arrays are views into the mesh)
        a = mesh.arrays()
        assert a["JxW"].min() > 0
        assert a["inv_jac"].shape[1:] == (3 ** dim, dim, dim)
        err.append(abs(a["JxW"].sum() / exact - 1.0))
    assert err[1] < 0.3 * err[0] and err[2] < 0.3 * err[1]  # second order in h
    # inv_jac really is the inverse of d x / d xi: check by differencing the quadrature-point coordinates of one cell
    mesh = mf.Mesh.ball(dim, 4, 1)
    a = mesh.arrays()
    _, _, xq, _, _ = o.shape_info(4)
    n = 5
    qp = a["quadrature_points"][3].reshape((n,) * dim + (dim,))
    Jinv = a["inv_jac"][3].reshape((n,) * dim + (dim, dim))
    # polynomial interpolation of x(xi) through the Gauss points is exact for a multilinear map: differentiate it
    D = np.array([o.lagrange_eval(xq, np.array([x]))[1][:, 0] for x in xq])  # D[q, t] = l_t'(x_q)
    for d in range(dim):
        ax = dim - 1 - d  # tensor direction d is array axis dim-1-d
        dx = np.moveaxis(np.tensordot(qp, D, axes=([ax], [1])), -1, ax)  # d x / d xi_d at the Gauss points
        F_col = dx  # [..., a] = d x_a / d xi_d
        # (J^-1 F)[e][d] = delta
        prod = np.einsum("...ea,...a->...e", Jinv, F_col)
        want = np.zeros(dim)
        want[d] = 1.0
        assert np.allclose(prod, want, atol=1e-11)


@pytest.mark.parametrize("dim,p,n_ref", [(2, 2, 1), (2, 4, 1), (3, 2, 1), (3, 3, 0)])
def test_oracle_three_way_differential_on_the_ball(dim, p, n_ref):
    mesh = mf.Mesh.ball(dim, p, n_ref)
    od = oracle_desc_from_mesh(mesh)
    assert not od.uniform_j0
    A = o.assemble(od)
    x = np.random.default_rng(dim * 10 + p).standard_normal(od.n_dofs)
    y = o.vmult(od, x)
    assert np.linalg.norm(y - A @ x) <= 1e-12 * np.linalg.norm(y)
    assert abs((A - A.T)).max() <= 1e-12 * abs(A).max()
    # constants are in the kernel of the unconstrained rows
    free = np.ones(od.n_dofs, bool)
    free[od.constrained] = False
    ones = np.ones(od.n_dofs)
    ones[od.constrained] = 0.0  # the constraint bracket zeroes them anyway
    full = o.Desc(dim, p, od.n_dofs, od.loc2glob, od.JxW, od.inv_jac, od.coefficient, np.zeros(0, np.uint32), None,
                  np.float64, od.shape_values, od.shape_gradients)
    assert np.abs(o.vmult(full, np.ones(od.n_dofs))).max() <= 1e-11 * abs(A).max()
