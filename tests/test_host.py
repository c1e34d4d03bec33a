"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the deal.II stand-in
mesh reproduces the oracle's independent mesh, the planner's invariants hold.  No GPU calls."""
import os
import re

import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from util import desc_from_oracle, emulate_plan_vmult, oracle_desc_from_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mfgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mfgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(mf.SYMBOLS)
    L = mf.lib()
    for s in declared:
        assert hasattr(L, s), s


def test_bad_arguments_fail_loudly():
    d = mf.Desc()
    with pytest.raises(mf.MfgpuError):
        mf.Plan(d)
    od = o.uniform_mesh_desc(2, 2, 2)
    bad = od.loc2glob.copy()
    bad[0, 0] = od.n_dofs  # out of range
    desc, keep = desc_from_oracle(od)
    desc2, keep2 = mf.make_desc(2, 2, od.n_dofs, bad, od.JxW, od.inv_jac, od.coefficient, od.constrained,
                                od.shape_values, od.shape_gradients)
    with pytest.raises(mf.MfgpuError, match="out of range"):
        mf.Plan(desc2, keep2)
    with pytest.raises(mf.MfgpuError):
        mf.Mesh.uniform(2, 9, 2)
    with pytest.raises(mf.MfgpuError):
        mf.Mesh.uniform(3, 2, [2, 2, 0])


@pytest.mark.parametrize("dim,p,n", [(2, 2, 32), (2, 4, 4), (3, 1, 3), (3, 4, 3), (3, 6, 2), (3, 3, [2, 3, 4])])
def test_mesh_standin_matches_oracle_mesh(dim, p, n):
    m = mf.Mesh.uniform(dim, p, n)
    od = o.uniform_mesh_desc(dim, p, n) if np.isscalar(n) else None
    a = m.arrays()
    if od is None:  # anisotropic counts: cube cells need equal h, oracle builder takes a tuple too
        od = o.uniform_mesh_desc(dim, p, tuple(n), lo=-1.0, hi=-1.0 + 2.0)
        # h = (hi-lo)/n[0] for both builders
    assert m.n_dofs == od.n_dofs and m.n_cells == od.n_cells
    np.testing.assert_array_equal(a["loc2glob"], od.loc2glob)
    np.testing.assert_array_equal(a["constrained_dofs"], od.constrained)
    np.testing.assert_allclose(a["JxW"], od.JxW, rtol=1e-13)
    np.testing.assert_allclose(a["inv_jac"], od.inv_jac, rtol=1e-15)
    np.testing.assert_allclose(a["shape_values"], od.shape_values, atol=1e-14)
    np.testing.assert_allclose(a["shape_gradients"], od.shape_gradients, atol=1e-12)
    np.testing.assert_allclose(a["constraint_weights"], od.weights, atol=1e-14)
    np.testing.assert_allclose(o.coefficient_value(a["quadrature_points"]), od.coefficient, rtol=1e-13)
    np.testing.assert_allclose(m.dof_coords(), od.dof_coords, atol=1e-14)


def test_c1_sizes():
    # BASELINE configs[0]: p=2, 2D, 5 global refinements = 32^2 cells, 65^2 dofs, 256 boundary dofs
    m = mf.Mesh.uniform(2, 2, 32)
    assert (m.n_cells, m.n_dofs, m.desc.n_constrained) == (1024, 4225, 256)


def _plan_invariants(od, plan, colored=False, planes=False):
    """planes: the plan of apply_planes3 (mfgpu_desc.kernel = PLANES, or the default at p = 4 in 3D): constrained dofs
    and the interior dofs beyond the dof list's interior slots take the pass-2 route although one batch touches them,
    every batch owns a fixed number of halo slots and at least one pass-2 dof."""
    nd = od.nd
    bco, bdo, cbo = plan.batch_cell_off, plan.batch_dof_off, plan.color_batch_off
    order, bdofs, bflags, lmap = plan.cell_order, plan.bdofs, plan.bflags, plan.lmap
    nb = len(bco) - 1
    # every cell exactly once
    assert sorted(order.tolist()) == list(range(od.n_cells))
    assert bco[0] == 0 and bco[-1] == od.n_cells and cbo[0] == 0 and cbo[-1] == nb
    if not colored:  # two-pass plans keep the spatial creation order: one launch over all batches
        assert len(cbo) == 2
    con = np.zeros(od.n_dofs, bool)
    con[od.constrained] = True
    touched = np.zeros(od.n_dofs, bool)
    ntouch = np.zeros(od.n_dofs, int)
    for b in range(nb):
        ntouch[bdofs[bdo[b]:bdo[b + 1]]] += 1
    # shared-dof CSR of the two-pass mode
    sd, so, si = plan.sdofs, plan.s_off, plan.s_idx
    sdg = sd & 0x7fffffff
    if planes:
        assert set(np.nonzero(ntouch >= 2)[0].tolist()) <= set(sdg.tolist()) and len(np.unique(sdg)) == len(sdg)
        assert (con[sdg[ntouch[sdg] == 1]] | True).all()
        assert set(np.nonzero(con & (ntouch >= 1))[0].tolist()) <= set(sdg.tolist())  # every touched constrained dof
        assert len(set(si.tolist())) == len(si) and (si < plan.halo_off[-1]).all()   # distinct halo slots
    else:
        np.testing.assert_array_equal(np.sort(sdg), np.nonzero(ntouch >= 2)[0])
        assert sorted(si.tolist()) == list(range(int(plan.halo_off[-1])))   # every halo slot read exactly once
    np.testing.assert_array_equal(np.diff(so), ntouch[sd & 0x7fffffff])
    on_route2 = np.zeros(od.n_dofs, bool)
    on_route2[sdg] = True
    # grouped form of the same pass: chunks of <= 64 consecutive shared dofs with one toucher set; partial t
    # of the chunk's lane l sits at gstarts[tstart + t] + offset + l -- must reproduce the CSR exactly
    ch, gs = plan.chunks, plan.gstarts
    covered = np.zeros(len(sd), int)
    for pos, ck, ts, off in ch.tolist():
        cnt, k = ck & 0xffff, ck >> 16
        assert 1 <= cnt <= 64 and (k >= 2 or planes)
        covered[pos:pos + cnt] += 1
        for lane in range(cnt):
            i = pos + lane
            assert so[i + 1] - so[i] == k
            np.testing.assert_array_equal(si[so[i]:so[i + 1]], gs[ts:ts + k] + off + lane)
    assert (covered == 1).all()
    for c in range(len(cbo) - 1):
        seen = np.zeros(od.n_dofs, bool)
        for b in range(cbo[c], cbo[c + 1]):
            g = bdofs[bdo[b]:bdo[b + 1]]
            f = bflags[bdo[b]:bdo[b + 1]]
            ni = int(plan.batch_nint[b])
            assert (np.diff(g[:ni].astype(np.int64)) > 0).all()     # [interior asc | shared, grouped], unique
            assert len(np.unique(g)) == len(g)
            assert (ntouch[g[:ni]] == 1).all() and not on_route2[g[:ni]].any() and on_route2[g[ni:]].all()
            if planes:
                assert len(g) > ni and not con[g[:ni]].any()
                assert plan.halo_off[b + 1] - plan.halo_off[b] >= len(g) - ni + 1
                assert plan.halo_off[b + 1] - plan.halo_off[b] == plan.halo_off[1] - plan.halo_off[0]
            else:
                assert (ntouch[g[ni:]] >= 2).all()
                assert plan.halo_off[b + 1] - plan.halo_off[b] == len(g) - ni
            assert not (colored and seen[g].any())                   # colour is conflict-free
            seen[g] = True
            np.testing.assert_array_equal((f & 1).astype(bool), con[g])
            np.testing.assert_array_equal(plan.bdofs_constrained[bdo[b]:bdo[b + 1]], con[g])
            np.testing.assert_array_equal((f & 2).astype(bool), touched[g])  # first toucher stores
            cells = np.arange(bco[b], bco[b + 1])
            np.testing.assert_array_equal(g[lmap[cells]], od.loc2glob[order[cells]])
            if not colored:
                touched[g] = True
        touched |= seen
    orph = plan.orphans
    np.testing.assert_array_equal(np.sort(orph & 0x7fffffff), np.nonzero(~touched)[0])
    np.testing.assert_array_equal((orph >> 31).astype(bool), con[orph & 0x7fffffff])


@pytest.mark.parametrize("dim,p,n,kw", [
    (2, 2, 8, {}), (3, 2, 4, {}), (3, 4, 3, {}), (3, 4, 6, dict(max_cells_per_batch=27)),
    (3, 1, 5, dict(max_cells_per_batch=8)), (2, 4, 6, dict(max_cells_per_batch=4, max_dofs_per_batch=60)),
    (3, 3, 3, dict(max_cells_per_batch=1)),
])
def test_plan_invariants_and_dataflow(dim, p, n, kw):
    od = o.uniform_mesh_desc(dim, p, n)
    rng = np.random.default_rng(0)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    for colored in (False, True):  # two-pass plan (spatial batch order) / coloured plan (colour-major order)
        desc, keep = desc_from_oracle(od, colored=colored, kernel=mf.KERNEL_PENCILS, **kw)
        plan = mf.Plan(desc, keep)
        _plan_invariants(od, plan, colored)
        ref = o.vmult(od, x)
        for tp in (False, True):  # the sequential emulation of either data flow is valid for both plans
            np.testing.assert_allclose(emulate_plan_vmult(od, plan, x, twopass=tp), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        ref = o.vmult_add(od, y0, x)
        for tp in (False, True):
            np.testing.assert_allclose(emulate_plan_vmult(od, plan, x, y0, twopass=tp), ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("p,n,kw", [(4, 3, {}), (4, 5, {}), (4, 4, dict(max_cells_per_batch=5)), (3, 5, {}), (2, 6, {}),
                                    (2, 4, dict(max_cells_per_batch=1))])
def test_plane_plan_invariants_and_dataflow(p, n, kw):
    """the plan apply_planes3 runs on (fixed slot structure of the dof lists, demotions to the pass-2 route, batches
    cut back to their surface bound): invariants, and the numpy emulation of its data flow against the oracle"""
    od = o.uniform_mesh_desc(3, p, n)
    rng = np.random.default_rng(0)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    desc, keep = desc_from_oracle(od, kernel=mf.KERNEL_PLANES, **kw)
    plan = mf.Plan(desc, keep)
    _plan_invariants(od, plan, False, planes=True)
    nbd, ni = np.diff(plan.batch_dof_off), plan.batch_nint
    ji, hs = {3: (2, 3), 4: (5, 6), 5: (9, 9)}[p + 1]  # p_ji, p_hs of mfgpu_internal.h
    assert (np.diff(plan.batch_cell_off) <= 64 // (p + 1)).all() and (ni <= 64 * ji).all() and (nbd - ni < 64 * hs).all()
    ref = o.vmult(od, x)
    np.testing.assert_allclose(emulate_plan_vmult(od, plan, x, twopass=True), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    ref = o.vmult_add(od, y0, x)
    np.testing.assert_allclose(emulate_plan_vmult(od, plan, x, y0, twopass=True), ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def _masks3():
    out = [t | (xyz << 3) for xyz in range(1, 8) for t in range(8)]
    for e in (1 << 6, 1 << 7, 1 << 8, (1 << 6) | (1 << 5), (1 << 7) | (1 << 3), (1 << 8) | (1 << 4)):
        out += [e | t for t in range(8)]
    return out


@pytest.mark.parametrize("p,n,kw", [(4, 3, {}), (4, 5, {}), (3, 4, {}), (2, 5, {}), (4, 4, dict(max_cells_per_batch=3)),
                                    (4, 7, {}), (3, 9, {})])  # (7 and 9 cells: box-seeded batches with leftover columns)
def test_plane_records_dataflow_with_hanging_node_batches(p, n, kw):
    """the fixed-size records apply_planes3 reads (dof lists, index runs, hanging-node records): numpy emulation of the
    kernel's data flow -- private entries, line-by-line interpolation passes with the plain weight matrix, plain cell
    kernel, transposed passes -- against the oracle's per-cell resolve_hanging_nodes, for every mask of the reference's
    known-answer test placed on cells of a conforming mesh; and on conforming meshes without masks"""
    from util import emulate_plane_records_vmult

    od = o.uniform_mesh_desc(3, p, n)
    rng = np.random.default_rng(p * 10 + n)
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    for masked in (False, True):
        if masked:
            masks = _masks3()
            cm = np.zeros(od.n_cells, dtype=np.uint32)
            pick = rng.permutation(od.n_cells)[:od.n_cells * 3 // 4]
            cm[pick] = np.array(masks, dtype=np.uint32)[np.arange(len(pick)) % len(masks)]
            od.constraint_mask = cm
        desc, keep = desc_from_oracle(od, kernel=mf.KERNEL_PLANES, **kw)
        plan = mf.Plan(desc, keep)
        assert (len(plan.pr_hn) > 0) == masked
        ref = o.vmult(od, x)
        np.testing.assert_allclose(emulate_plane_records_vmult(od, plan, x), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        ref = o.vmult_add(od, y0, x)
        np.testing.assert_allclose(emulate_plane_records_vmult(od, plan, x, y0), ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def test_plane_records_on_the_adaptive_mesh():
    from util import emulate_plane_records_vmult, oracle_desc_from_mesh

    mesh = mf.Mesh.adaptive(3, 4, 4)
    od = oracle_desc_from_mesh(mesh)
    plan = mf.Plan(mesh.desc, mesh)
    assert len(plan.pr_hn) > 0
    # batches of masked and of unmasked cells stay interleaved in creation order: ONE launch walks them all
    masked = plan.pr_hn_slot != 0xffffffff
    assert masked.any() and (~masked).any() and np.any(masked[:-1] & ~masked[1:]) and np.any(~masked[:-1] & masked[1:])
    cm = mesh.arrays()["constraint_mask"][plan.cell_order]
    bco = plan.batch_cell_off
    for b in range(len(masked)):
        assert np.all((cm[bco[b]:bco[b + 1]] != 0) == masked[b])  # a batch holds one kind of cells only
    x = np.random.default_rng(1).standard_normal(od.n_dofs)
    ref = o.vmult(od, x)
    np.testing.assert_allclose(emulate_plane_records_vmult(od, plan, x), ref, rtol=0, atol=1e-12 * np.abs(ref).max())


def test_plan_orphans_and_ragged_mesh():
    """dofs no cell touches (as hanging nodes are after substitution) + a disconnected, ragged mesh"""
    od = o.uniform_mesh_desc(2, 2, 4)
    keep_cells = np.array([0, 1, 2, 5, 10, 15])  # holes -> orphans, disconnected pieces
    od2 = o.Desc(2, 2, od.n_dofs, od.loc2glob[keep_cells], od.JxW[keep_cells], od.inv_jac[keep_cells],
                 od.coefficient[keep_cells], od.constrained)
    x = np.random.default_rng(1).standard_normal(od.n_dofs)
    ref = o.vmult(od2, x)
    for colored in (False, True):
        desc, keep = desc_from_oracle(od2, max_cells_per_batch=3, colored=colored)
        plan = mf.Plan(desc, keep)
        _plan_invariants(od2, plan, colored)
        assert len(plan.orphans) > 0
        for tp in (False, True):
            np.testing.assert_allclose(emulate_plan_vmult(od2, plan, x, twopass=tp), ref, atol=1e-12 * np.abs(ref).max())


def test_batching_quality_structured():
    """greedy batching should find compact 3x3x3 blocks on a structured p=4 mesh (pencil kernels), and exact
    3x2x2 boxes, long side along x, for the plane kernel (12 cells = one wave)"""
    od = o.uniform_mesh_desc(3, 4, 9)
    desc, keep = desc_from_oracle(od, max_cells_per_batch=27, kernel=mf.KERNEL_PENCILS_X)
    plan = mf.Plan(desc, keep)
    nb = len(plan.batch_cell_off) - 1
    assert nb == 27, nb
    assert np.diff(plan.batch_dof_off).max() == 13 ** 3
    assert plan.batch_nint.min() == 11 ** 3  # the one batch in the middle of the 3x3x3 arrangement
    # two-pass plans run the batches in spatial (creation) order: consecutive batches are x-neighbours,
    # which is what lets concurrently running workgroups share src lines in one XCD's L2
    first = plan.cell_order[plan.batch_cell_off[:-1]]
    np.testing.assert_array_equal(first, [3 * bx + 9 * (3 * by) + 81 * (3 * bz)
                                          for bz in range(3) for by in range(3) for bx in range(3)])
    desc, keep = desc_from_oracle(od, max_cells_per_batch=27, colored=True)
    assert len(mf.Plan(desc, keep).color_batch_off) - 1 == 8
    od = o.uniform_mesh_desc(3, 4, 12)
    desc, keep = desc_from_oracle(od)  # the default at p = 4: apply_planes3
    plan = mf.Plan(desc, keep)
    assert len(plan.batch_cell_off) - 1 == 12 ** 3 // 12
    for b in range(len(plan.batch_cell_off) - 1):
        cells = plan.cell_order[plan.batch_cell_off[b]:plan.batch_cell_off[b + 1]].astype(np.int64)
        ext = [int(np.ptp(cells // 12 ** d % 12)) + 1 for d in range(3)]
        assert ext == [3, 2, 2], ext
    assert np.diff(plan.batch_dof_off).max() == 13 * 9 * 9 and plan.batch_nint.max() == 11 * 7 * 7
    # a size the box does not divide: the batches start as boxes (face neighbours by direction from the dof lists), so
    # the 16th column is batched by itself instead of unravelling the whole mesh into 9-11-cell batches
    od = o.uniform_mesh_desc(3, 4, 16)
    desc, keep = desc_from_oracle(od)
    plan = mf.Plan(desc, keep)
    sizes = np.diff(plan.batch_cell_off)
    assert len(sizes) <= 355 and (sizes == 12).sum() >= 300, (len(sizes), np.bincount(sizes))  # (greedy alone: 374 / 171; ideal 341.3)
    assert sorted(plan.cell_order.tolist()) == list(range(16 ** 3))
    # p = 6: nine cells per wave, 3x3x1 slabs where the mesh allows
    od = o.uniform_mesh_desc(3, 6, 6)
    desc, keep = desc_from_oracle(od)
    plan = mf.Plan(desc, keep)
    assert len(plan.batch_cell_off) - 1 == 6 ** 3 // 9


def test_slab_meshes_tile_the_global_mesh():
    """multi-GPU partition: z-slabs with consistent interface planes"""
    dim, p, n = 3, 2, 4
    full = mf.Mesh.uniform(dim, p, n)
    lo_m = mf.Mesh.uniform(dim, p, n, slab=(0, 2))
    hi_m = mf.Mesh.uniform(dim, p, n, slab=(2, 4))
    assert lo_m.n_cells + hi_m.n_cells == full.n_cells
    i_up, i_lo = lo_m.interface_dofs(1), hi_m.interface_dofs(0)
    assert len(i_up) == len(i_lo) == (p * n + 1) ** 2
    assert len(lo_m.interface_dofs(0)) == 0 and len(hi_m.interface_dofs(1)) == 0
    np.testing.assert_allclose(lo_m.dof_coords()[i_up], hi_m.dof_coords()[i_lo], atol=1e-15)
    # interface dofs that are on the global boundary are constrained on both sides
    c_lo = np.isin(i_up, lo_m.arrays()["constrained_dofs"])
    c_hi = np.isin(i_lo, hi_m.arrays()["constrained_dofs"])
    np.testing.assert_array_equal(c_lo, c_hi)
    assert c_lo.sum() == (p * n + 1) ** 2 - (p * n - 1) ** 2
