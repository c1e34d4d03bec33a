"""Host-side tests of the optional dof renumbering (mfgpu_suggest_renumbering / mfgpu_mesh_renumber; deal.II's
MatrixFree::renumber_dofs + DoFHandler::renumber_dofs): it is a permutation, it leaves the plan's batches alone, a
batch's own dofs become one contiguous run, and the renumbered description defines the same operator."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from util import emulate_plan_vmult, oracle_desc_from_mesh


@pytest.mark.parametrize("make,kern", [(lambda: mf.Mesh.uniform(3, 4, 6), 0), (lambda: mf.Mesh.uniform(3, 2, 8), 0),
                                       (lambda: mf.Mesh.uniform(2, 3, 9), 0), (lambda: mf.Mesh.adaptive(3, 2, 4), 0),
                                       (lambda: mf.Mesh.uniform(3, 3, 6), mf.KERNEL_PLANES)])
def test_renumbering_is_a_permutation_that_keeps_the_plan_and_the_operator(make, kern):
    mesh = make()
    mesh.desc.kernel = kern
    mesh.desc.max_cells_per_batch = 8 if mesh.desc.degree != 4 else 0
    od0 = oracle_desc_from_mesh(mesh)
    plan0 = mf.Plan(mesh.desc, mesh)
    order0, bco0 = plan0.cell_order, plan0.batch_cell_off
    ni = mesh.suggest_renumbering()
    assert np.array_equal(np.sort(ni), np.arange(mesh.n_dofs))
    x0 = np.random.default_rng(0).standard_normal(mesh.n_dofs)
    y0 = o.vmult(od0, x0)
    xc0 = mesh.dof_coords().copy()
    mesh.renumber(ni)
    mesh.desc.kernel = kern
    mesh.desc.max_cells_per_batch = 8 if mesh.desc.degree != 4 else 0
    od1 = oracle_desc_from_mesh(mesh)
    x1 = np.empty_like(x0)
    x1[ni] = x0
    y1 = o.vmult(od1, x1)
    np.testing.assert_allclose(y1[ni], y0, rtol=0, atol=1e-12 * np.abs(y0).max())
    np.testing.assert_array_equal(mesh.dof_coords()[ni], xc0)
    c = mesh.arrays()["constrained_dofs"]
    assert np.all(np.diff(c.astype(np.int64)) > 0)
    plan1 = mf.Plan(mesh.desc, mesh)
    np.testing.assert_array_equal(plan1.cell_order, order0)
    np.testing.assert_array_equal(plan1.batch_cell_off, bco0)
    # a batch's own dofs: one contiguous run each, batch after batch
    bdo, nint, bd = plan1.batch_dof_off, plan1.batch_nint, plan1.bdofs
    nxt = 0
    for b in range(len(bdo) - 1):
        own = bd[bdo[b]:bdo[b] + nint[b]].astype(np.int64)
        assert np.array_equal(own, np.arange(nxt, nxt + len(own)))
        nxt += len(own)
    ref = o.vmult(od1, x1)
    np.testing.assert_allclose(emulate_plan_vmult(od1, plan1, x1, twopass=True), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    with pytest.raises(mf.MfgpuError):
        mesh.renumber(np.zeros(mesh.n_dofs, np.uint32))
