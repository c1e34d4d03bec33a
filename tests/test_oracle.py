"""The oracle against everything the reference holds for this path:
published 1D tables, the three-way differential of test_laplace_op.cu, the hanging-node
interpolation known-answer test of test_hanging_node_interpolation.cu, analytic properties."""
import numpy as np
import pytest

from oracle import mf_oracle as o


def test_published_1d_tables():
    # reference 1-D tables for p = 4
    np.testing.assert_allclose(o.gll_01(4), [0, 0.1726731646, 0.5, 0.8273268354, 1], atol=2e-10)
    x, w = o.gauss_01(5)
    np.testing.assert_allclose(x, [0.0469100770, 0.2307653449, 0.5, 0.7692346551, 0.9530899230], atol=2e-10)
    np.testing.assert_allclose(w, [0.1184634425, 0.2393143352, 0.2844444444, 0.2393143352, 0.1184634425], atol=2e-10)
    # known-answer hanging-node weight tables (FE_Q subface interpolation, child 0)
    np.testing.assert_allclose(o.constraint_weights(1).reshape(2, 2), [[1, 0], [0.5, 0.5]], atol=1e-15)
    np.testing.assert_allclose(o.constraint_weights(2).reshape(3, 3),
                               [[1, 0, 0], [0.375, 0.75, -0.125], [0, 1, 0]], atol=1e-15)


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6])
def test_shape_tables_partition_of_unity(p):
    sv, sg, xq, wq, nodes = o.shape_info(p)
    n = p + 1
    np.testing.assert_allclose(sv.reshape(n, n).sum(0), 1.0, atol=1e-13)
    np.testing.assert_allclose(sg.reshape(n, n).sum(0), 0.0, atol=1e-11)
    np.testing.assert_allclose(wq.sum(), 1.0, atol=1e-14)
    # derivative of x interpolated: sum_i x_i phi_i'(x_q) = 1
    np.testing.assert_allclose(nodes @ sg.reshape(n, n), 1.0, atol=1e-11)


def test_three_way_differential_reference_config():
    """test_laplace_op.cu:47-48,136-137: dim 2, p 4, hyper_cube(0,1), refine_global(2) = 16 cells,
    289 dofs, Dirichlet rows -> identity; matrix-free apply == assembled SparseMatrix apply."""
    d = o.uniform_mesh_desc(2, 4, 4, 0.0, 1.0)
    assert d.n_dofs == 289 and d.n_cells == 16
    x = np.random.default_rng(1).random(d.n_dofs)  # reference: rand()/RAND_MAX, unseeded
    y_mf = o.vmult(d, x)
    y_sp = o.assemble(d) @ x
    assert np.linalg.norm(y_mf - y_sp) <= 1e-12 * np.linalg.norm(y_sp)


@pytest.mark.parametrize("dim,p,n", [(2, 1, 5), (2, 2, 4), (2, 3, 3), (3, 1, 3), (3, 2, 3), (3, 4, 2), (3, 6, 2)])
def test_sumfac_equals_assembled(dim, p, n):
    d = o.uniform_mesh_desc(dim, p, n)
    x = np.random.default_rng(dim * 10 + p).standard_normal(d.n_dofs)
    y_mf = o.vmult(d, x)
    A = o.assemble(d)
    y_sp = A @ x
    assert np.linalg.norm(y_mf - y_sp) <= 1e-12 * np.linalg.norm(y_sp)
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()


def test_float_tolerance():
    d64 = o.uniform_mesh_desc(3, 2, 3)
    d32 = o.uniform_mesh_desc(3, 2, 3, dtype=np.float32)
    x = np.random.default_rng(5).standard_normal(d64.n_dofs)
    y64 = o.vmult(d64, x)
    y32 = o.vmult(d32, x.astype(np.float32))
    assert y32.dtype == np.float32
    assert np.linalg.norm(y32 - y64) <= 1e-5 * np.linalg.norm(y64)


def test_analytic_properties():
    d = o.uniform_mesh_desc(3, 3, 3)
    rng = np.random.default_rng(2)
    u, v = rng.standard_normal(d.n_dofs), rng.standard_normal(d.n_dofs)
    # symmetry
    assert abs(v @ o.vmult(d, u) - u @ o.vmult(d, v)) <= 1e-11 * abs(v @ o.vmult(d, u))
    # positive semi-definite
    assert u @ o.vmult(d, u) > 0
    # constants are in the kernel away from the boundary: rows whose cells touch no Dirichlet dof
    y = o.vmult(d, np.ones(d.n_dofs))
    con = np.zeros(d.n_dofs, bool)
    con[d.constrained] = True
    touches = np.zeros(d.n_dofs, bool)
    for c in range(d.n_cells):
        if con[d.loc2glob[c]].any():
            touches[d.loc2glob[c]] = True
    assert np.abs(y[~touches]).max() <= 1e-11
    np.testing.assert_allclose(y[d.constrained], 1.0)  # identity rows (laplace_operator_gpu.h:300-302)
    # a == 1, u = x interpolated, no constraints: u^T K u = |Omega|
    d1 = o.uniform_mesh_desc(3, 2, 2, coefficient=lambda x: np.ones(x.shape[:-1]))
    d1.constrained = np.zeros(0, dtype=np.uint32)
    ux = d1.dof_coords[:, 0]
    assert abs(ux @ o.vmult(d1, ux) - 8.0) <= 1e-11


def test_vmult_add_semantics():
    d = o.uniform_mesh_desc(2, 2, 3)
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(d.n_dofs), rng.standard_normal(d.n_dofs)
    y = o.vmult_add(d, y0, x)
    ref = y0 + o.vmult(d, x)
    # on constrained rows the reference gives dst_in + src (constraint_handler_gpu.cu:286)
    np.testing.assert_allclose(y, ref, atol=1e-13)


def test_bmop_protocol_first_apply():
    """C1 (bmop-cpu.cc defaults scaled down): src == 0.1 -> boundary rows stay 0.1, interior rows whose
    cells touch no boundary are 0 after the first apply (grad const = 0)."""
    d = o.uniform_mesh_desc(2, 2, 8)
    y = o.bmop_protocol(d, 1)
    np.testing.assert_allclose(y[d.constrained], 0.1)
    gi = np.round((d.dof_coords + 1) / 2 * 16).astype(int)
    deep = ((gi >= 3) & (gi <= 13)).all(1)
    assert np.abs(y[deep]).max() < 1e-12


# ---- hanging-node interpolation known-answer test -------------------------------------------

def _foo(x, y, z):
    return x + 2 * y + 3 * z  # test_hanging_node_interpolation.cu:76-82


def _setup_values(vec, mask, dim, p):
    """test_hanging_node_interpolation.cu:85-207 (setup_values): overwrite constrained faces with the
    coarse neighbour's values."""
    n = p + 1
    nodes = o.gll_01(p)
    TX, TY, TZ, FX, FY, FZ = 1, 2, 4, 8, 16, 32
    if dim == 2:
        if mask & FX:
            j = 0 if mask & TX else p
            for i in range(n):
                x, y = nodes[j], 2 * nodes[i]
                if not mask & TY:
                    y -= 1.0
                vec[i, j] = _foo(x, y, 1)
        if mask & FY:
            i = 0 if mask & TY else p
            for j in range(n):
                x, y = 2 * nodes[j], nodes[i]
                if not mask & TX:
                    x -= 1.0
                vec[i, j] = _foo(x, y, 1)
        return
    if mask & FX:
        k = 0 if mask & TX else p
        for i in range(n):
            for j in range(n):
                x, y, z = nodes[k], 2 * nodes[j], 2 * nodes[i]
                if not mask & TY:
                    y -= 1.0
                if not mask & TZ:
                    z -= 1.0
                vec[i, j, k] = _foo(x, y, z)
    if mask & FY:
        j = 0 if mask & TY else p
        for i in range(n):
            for k in range(n):
                x, y, z = 2 * nodes[k], nodes[j], 2 * nodes[i]
                if not mask & TX:
                    x -= 1.0
                if not mask & TZ:
                    z -= 1.0
                vec[i, j, k] = _foo(x, y, z)
    if mask & FZ:
        i = 0 if mask & TZ else p
        for j in range(n):
            for k in range(n):
                x, y, z = 2 * nodes[k], 2 * nodes[j], nodes[i]
                if not mask & TX:
                    x -= 1.0
                if not mask & TY:
                    y -= 1.0
                vec[i, j, k] = _foo(x, y, z)


def hn_kat_cases(dim, p):
    """all masks of loop_over_constraints<dim,p>(combinations=true), :311-350"""
    n = p + 1
    nodes = o.gll_01(p)
    if dim == 2:
        Y, X = np.meshgrid(nodes, nodes, indexing="ij")
        ref = _foo(X, Y, 1.0)
    else:
        Z, Y, X = np.meshgrid(nodes, nodes, nodes, indexing="ij")
        ref = _foo(X, Y, Z)
    for xyz in range((1 << dim) - 1):
        for t in range(1 << dim):
            mask = t | (xyz << 3)
            vec = ref.copy()
            _setup_values(vec, mask, dim, p)
            yield mask, vec, ref


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_hanging_node_interpolation_kat(dim, p):
    W = o.constraint_weights(p)
    count = 0
    for mask, vec, ref in hn_kat_cases(dim, p):
        out = o.hn_resolve(vec, mask, dim, p, W, False)
        tol = 0.0 if p == 1 else 1e-14  # reference expects exactly 0; GLL weights round at 1e-15
        assert np.abs(out - ref).max() <= tol, (dim, p, mask)
        count += 1
    assert count == (12 if dim == 2 else 56)


@pytest.mark.parametrize("dim,p", [(2, 3), (3, 2), (3, 4)])
def test_hanging_node_transpose_is_adjoint(dim, p):
    W = o.constraint_weights(p)
    rng = np.random.default_rng(7)
    n = p + 1
    masks = [m for m, _, _ in hn_kat_cases(dim, p)]
    if dim == 3:
        masks += [1 << 6, (1 << 7) | 3, (1 << 8) | 5, (1 << 6) | (1 << 5) | 7]  # edge masks
    for mask in masks:
        u = rng.standard_normal((n,) * dim)
        v = rng.standard_normal((n,) * dim)
        Cu = o.hn_resolve(u, mask, dim, p, W, False)
        Ctv = o.hn_resolve(v, mask, dim, p, W, True)
        assert abs((Cu * v).sum() - (u * Ctv).sum()) <= 1e-13 * (np.abs(Cu * v).sum() + 1e-30)
