"""GPU parity of the level operator with refinement edges (laplace_operator_gpu.h:154-186, 306-352):
level matrix with Dirichlet + edge dofs constrained, and the interface matrices vmult_interface_down / up that deal.II's
Multigrid takes as edge matrices, against the oracle's restatement of the reference's bracketed cell loops.  The edge
sets are synthetic (the dofs on the surface of a box of cells inside a uniform level mesh: what the refinement edge of
a locally refined patch looks like from the level below).  Tolerance: relative l2 <= 1e-12 (double), 1e-5 (float)."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from test_gpu import TOL, rel
from util import oracle_desc_from_mesh

pytestmark = pytest.mark.gpu


def _edge_of_box(mesh, lo, hi):
    """dofs on the surface of the box [lo, hi]^dim (dof coordinates), excluding nothing: a closed interface"""
    x = mesh.dof_coords()
    inside = np.all((x >= lo - 1e-12) & (x <= hi + 1e-12), axis=1)
    on = inside & np.any((np.abs(x - lo) < 1e-12) | (np.abs(x - hi) < 1e-12), axis=1)
    return np.nonzero(on)[0].astype(np.uint32)


@pytest.mark.parametrize("dim,p,n,nt", [(2, 2, 8, mf.F64), (2, 4, 6, mf.F64), (3, 1, 6, mf.F64), (3, 2, 4, mf.F64), (3, 4, 4, mf.F64),
                                        (3, 4, 6, mf.F32), (3, 3, 4, mf.F64)])
@pytest.mark.parametrize("touch_boundary", [False, True])
def test_level_operator_and_interface_matrices(dim, p, n, nt, touch_boundary):
    mesh = mf.Mesh.uniform(dim, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    h = 2.0 / n
    # touch_boundary: the box starts AT the domain boundary, so some edge dofs are Dirichlet dofs too (a refinement
    # edge that reaches the boundary, as mfgpu_mg_hierarchy produces them)
    edge = _edge_of_box(mesh, -1.0 + (0.0 if touch_boundary else h), -1.0 + h * (n // 2 + 1))
    if touch_boundary:
        assert len(np.intersect1d(edge, od.constrained)) > 0
    assert 0 < len(edge) < mesh.n_dofs
    lev = mf.Level(mesh.desc, edge, mesh)
    rng = np.random.default_rng(dim * 10 + p)
    x = rng.standard_normal(mesh.n_dofs).astype(mf.np_dtype(nt)).astype(np.float64)
    a, b = mf.DeviceVector(mesh.n_dofs, nt), mf.DeviceVector(mesh.n_dofs, nt)
    a.from_host(x)
    # level matrix: identity rows on Dirichlet AND edge dofs
    odc = o.Desc(dim, p, od.n_dofs, od.loc2glob, od.JxW, od.inv_jac, od.coefficient,
                 np.union1d(od.constrained, edge).astype(np.uint32), None, np.float64, od.shape_values, od.shape_gradients)
    lev.vmult(b, a)
    mf.synchronize()
    assert rel(b.to_host(), o.vmult(odc, x)) <= TOL[nt]
    for fn, ref in ((lev.vmult_interface_down, o.vmult_interface_down), (lev.vmult_interface_up, o.vmult_interface_up)):
        b.fill(5.0)
        fn(b, a)
        mf.synchronize()
        want = ref(od, edge, x)
        got = b.to_host().astype(np.float64)
        assert np.linalg.norm(got - want) <= TOL[nt] * np.linalg.norm(want)
        np.testing.assert_array_equal(a.to_host(), x.astype(mf.np_dtype(nt)))  # src untouched
    # the two are transposes of each other: <down x, y> = <x, up y>
    if nt == mf.F64:
        y = rng.standard_normal(mesh.n_dofs)
        assert abs(o.vmult_interface_down(od, edge, x) @ y - x @ o.vmult_interface_up(od, edge, y)) <= 1e-10 * np.linalg.norm(x) * np.linalg.norm(y)


def test_level_without_edges_and_errors():
    mesh = mf.Mesh.uniform(3, 2, 3)
    lev = mf.Level(mesh.desc, np.zeros(0, np.uint32), mesh)
    a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
    a.fill(1.0)
    b.fill(3.0)
    lev.vmult_interface_down(b, a)
    mf.synchronize()
    assert not b.to_host().any()  # no refinement edge: the interface matrices are zero
    with pytest.raises(mf.MfgpuError):
        mf.Level(mesh.desc, np.array([mesh.n_dofs], np.uint32), mesh)
    am = mf.Mesh.adaptive(3, 2, 4)
    with pytest.raises(mf.MfgpuError):
        mf.Level(am.desc, np.zeros(0, np.uint32), am)
