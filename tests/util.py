"""Shared helpers for the tests (host side only)."""
import numpy as np

import pymfgpu as mf
from oracle import mf_oracle as o


def oracle_desc_from_mesh(mesh: "mf.Mesh", dtype=None) -> o.Desc:
    """Oracle description built from the SAME plain arrays the product hands to mfgpu_create."""
    a = mesh.arrays()
    d = mesh.desc
    dt = mf.np_dtype(d.number_type) if dtype is None else dtype
    coef = o.coefficient_value(a["quadrature_points"].astype(np.float64))
    # The 1D tables and hanging-node weights are the ORACLE's own (oracle/mf_oracle.py shape_info / constraint_weights,
    # checked there against the published GLL / Gauss values), not the product's: the product's copies are compared
    # with them here, so a wrong table in the mesh stand-in cannot cancel out of a GPU-vs-oracle comparison.
    sv, sg, _, _, _ = o.shape_info(d.degree)
    w = o.constraint_weights(d.degree)
    tol = 1e-13 if d.number_type == mf.F64 else 1e-6
    assert np.abs(a["shape_values"].astype(np.float64).reshape(sv.shape) - sv).max() <= tol
    assert np.abs(a["shape_gradients"].astype(np.float64).reshape(sg.shape) - sg).max() <= tol * max(1.0, np.abs(sg).max())
    assert np.abs(a["constraint_weights"].reshape(w.shape) - w).max() <= 1e-13
    return o.Desc(d.dim, d.degree, d.n_dofs, a["loc2glob"], a["JxW"], a["inv_jac"], coef,
                  a["constrained_dofs"], a["constraint_mask"], dt, sv, sg, w, mesh.dof_coords())


def desc_from_oracle(od: o.Desc, number_type=mf.F64, **kw):
    """C-ABI description from an oracle Desc (independent oracle-side mesh -> product)."""
    return mf.make_desc(od.dim, od.degree, od.n_dofs, od.loc2glob, od.JxW, od.inv_jac, od.coefficient,
                        od.constrained, od.shape_values, od.shape_gradients, number_type,
                        od.constraint_mask, od.weights if od.constraint_mask is not None else None, **kw)


def emulate_plan_vmult(od: o.Desc, plan: "mf.Plan", src, dst_in=None, twopass=False):
    """numpy emulation of the kernel's DATA FLOW driven by the plan arrays (gather through
    bdofs/lmap, per-batch accumulation, first-touch store / add, orphans), with the oracle as the
    cell kernel.  Checks the planner, not the HIP code."""
    add = dst_in is not None
    dst = np.full(od.n_dofs, np.nan) if not add else np.array(dst_in, dtype=np.float64)
    bco, bdo, order = plan.batch_cell_off, plan.batch_dof_off, plan.cell_order
    bdofs, bflags, lmap = plan.bdofs, plan.bflags, plan.lmap
    nint, hoff = plan.batch_nint, plan.halo_off
    halo = np.full(int(hoff[-1]), np.nan)
    # permuted oracle desc (plan cell order)
    pd = o.Desc(od.dim, od.degree, od.n_dofs, od.loc2glob[order], od.JxW[order], od.inv_jac[order],
                od.coefficient[order], od.constrained,
                None if od.constraint_mask is None else od.constraint_mask[order], od.dtype,
                od.shape_values, od.shape_gradients, od.weights)
    for b in range(len(bco) - 1):
        g = bdofs[bdo[b]:bdo[b + 1]]
        f = bflags[bdo[b]:bdo[b + 1]]
        usrc = np.where(f & 1, 0.0, src[g])
        acc = np.zeros(len(g))
        cells = np.arange(bco[b], bco[b + 1])
        sub = o.Desc(pd.dim, pd.degree, pd.n_dofs, pd.loc2glob[cells], pd.JxW[cells], pd.inv_jac[cells],
                     pd.coefficient[cells], pd.constrained,
                     None if pd.constraint_mask is None else pd.constraint_mask[cells], pd.dtype,
                     pd.shape_values, pd.shape_gradients, pd.weights)
        loc = o.cell_apply(sub, usrc[lmap[cells]])
        np.add.at(acc, lmap[cells].reshape(-1), loc.reshape(-1))
        con, addf = (f & 1).astype(bool), (f & 2).astype(bool)
        if twopass:
            ni = int(nint[b])
            gi, ci = g[:ni], con[:ni]
            val = np.where(ci, src[gi], acc[:ni])
            dst[gi] = (dst[gi] if add else 0.0) + val
            halo[hoff[b]:hoff[b] + len(acc) - ni] = acc[ni:]  # (plane plans: hoff[b + 1] - hoff[b] is a fixed stride)
            continue
        first_con = con & ~addf
        dst[g[first_con]] = (dst[g[first_con]] if add else 0.0) + src[g[first_con]]
        free_add = ~con & (addf | add)
        dst[g[free_add]] += acc[free_add]
        free_store = ~con & ~(addf | add)
        dst[g[free_store]] = acc[free_store]
    if twopass:
        sd, so, si = plan.sdofs, plan.s_off, plan.s_idx
        for i in range(len(sd)):
            g = int(sd[i] & 0x7fffffff)
            val = src[g] if (sd[i] >> 31) else sum(halo[si[so[i]:so[i + 1]]])
            dst[g] = (dst[g] if add else 0.0) + val
    for oo in plan.orphans:
        g = int(oo & 0x7fffffff)
        s = src[g] if (oo >> 31) else 0.0
        dst[g] = (dst[g] if add else 0.0) + s
    return dst


def deform(od: o.Desc, eps=0.12, seed=0):
    """General-geometry version of a Cartesian oracle Desc (any mesh, hanging-node masks kept): per quadrature
    point F = h_cell (I + eps R), R random in [-1,1]; inv_jac = F^-1, JxW scaled by det(F) / h^dim."""
    assert od.uniform_j0
    nc, nd, dim = od.n_cells, od.nd, od.dim
    h = (1.0 / od.inv_jac.astype(np.float64)).reshape(nc, 1, 1, 1)
    F = h * (np.eye(dim) + eps * np.random.default_rng(seed).uniform(-1.0, 1.0, (nc, nd, dim, dim)))
    det = np.linalg.det(F)
    assert det.min() > 0
    jxw = od.JxW.astype(np.float64).reshape(nc, nd) / h.reshape(nc, 1) ** dim * det
    return o.Desc(dim, od.degree, od.n_dofs, od.loc2glob, jxw, np.linalg.inv(F), od.coefficient, od.constrained,
                  od.constraint_mask, od.dtype, od.shape_values, od.shape_gradients, od.weights)


def deformed_oracle_desc(p, n, eps=0.12, seed=0, dtype=np.float64):
    """3D oracle Desc with GENERAL geometry data: the uniform cube's connectivity with a
    synthetic reference->physical Jacobian F = h (I + eps R) per quadrature point (R random in [-1,1]), i.e.
    inv_jac = F^-1 [cell][q][d1][d2] and JxW = det(F) w_q.  The operator is defined by these arrays alone
    (fee_gpu.cuh:235-241,275-281), so any smooth or rough field with det F > 0 exercises the code path."""
    od = o.uniform_mesh_desc(3, p, n, dtype=dtype)
    nc, nd = od.n_cells, od.nd
    h = 2.0 / n
    rng = np.random.default_rng(seed)
    F = h * (np.eye(3) + eps * rng.uniform(-1.0, 1.0, (nc, nd, 3, 3)))
    det = np.linalg.det(F)
    assert det.min() > 0
    jxw = od.JxW.astype(np.float64).reshape(nc, nd) / h ** 3 * det
    return o.Desc(3, p, od.n_dofs, od.loc2glob, jxw, np.linalg.inv(F), od.coefficient, od.constrained, None, dtype,
                  od.shape_values, od.shape_gradients)


def emulate_plane_records_vmult(od: o.Desc, plan: "mf.Plan", src, dst_in=None):
    """numpy emulation of apply_planes3's DATA FLOW on its fixed-size records (mfgpu_plan.cpp build_plane_records): dof
    lists, index runs and -- for the batches of cells with a hanging-node mask -- the hanging-node records: private
    copies of the constrained nodes, the interpolation passes x, y, z as plain W mat-vecs on the listed lines, the
    PLAIN cell kernel (the oracle without masks), the transposed passes in reverse order, the sums back to the dofs'
    entries; then the scatter and pass 2.  Checks the record builder (and hn_cell_lines), not the HIP code."""
    n, nd = od.n, od.nd
    n2 = n * n
    JS, HS, PRIV = {3: (2, 3, 192), 4: (5, 6, 256), 5: (9, 9, 384)}[n]  # p_ji, p_hs, p_priv_max of mfgpu_internal.h
    KGU = JS + HS
    JI, NB, NT, NIW = JS * 64, KGU * 64, (64 // n) * n, (n2 + 1) // 2
    CR, OR = PRIV // 64, 2
    HROWS = CR + 3 * OR * 3 + 2
    bco, order = plan.batch_cell_off, plan.cell_order
    nbat = len(bco) - 1
    bd = plan.pr_dofs.reshape(nbat, NB)
    ixw = plan.pr_idx.reshape(nbat, NIW, NT)
    hn = plan.pr_hn.reshape(-1, HROWS, 64)
    hn_slot = plan.pr_hn_slot  # batches of masked and of unmasked cells are interleaved (one launch)
    assert len(hn_slot) == nbat and sorted(hn_slot[hn_slot != 0xffffffff]) == list(range(hn.shape[0]))
    W = o.constraint_weights(od.degree).reshape(n, n)
    hoff = plan.halo_off
    add = dst_in is not None
    dst = np.full(od.n_dofs, np.nan) if not add else np.array(dst_in, dtype=np.float64)
    halo = np.full(int(hoff[-1]) + 1, np.nan)
    halo[-1] = 0.0
    pd = o.Desc(od.dim, od.degree, od.n_dofs, od.loc2glob[order], od.JxW[order], od.inv_jac[order],
                od.coefficient[order], od.constrained, None, od.dtype, od.shape_values, od.shape_gradients, od.weights)
    cm = None if od.constraint_mask is None else od.constraint_mask[order]
    for b in range(nbat):
        e = bd[b]
        g = (e & 0x7fffffff).astype(np.int64)
        con = (e >> 31).astype(bool)
        ua = np.zeros(NB + PRIV)
        ua[:NB] = np.where(con, 0.0, src[g])
        copies, ops = [], [[], [], []]
        if hn_slot[b] != 0xffffffff:
            rec = hn[hn_slot[b]]
            ncopy, cnt = int(rec[HROWS - 2, 0] & 0xffff), [int(rec[HROWS - 2, 0] >> 16), int(rec[HROWS - 1, 0] & 0xffff), int(rec[HROWS - 1, 0] >> 16)]
            assert np.all(rec[HROWS - 2] == rec[HROWS - 2, 0]) and np.all(rec[HROWS - 1] == rec[HROWS - 1, 0])
            flat = rec[:CR].reshape(-1)[:ncopy]
            copies = [(int(w >> 16), int(w & 0xffff)) for w in flat]
            for d in range(3):
                for k in range(cnt[d]):
                    ww = rec[CR + (d * OR + k // 64) * 3:CR + (d * OR + k // 64) * 3 + 3, k % 64]
                    ops[d].append([int((ww[t >> 1] >> (16 * (t & 1))) & 0xffff) for t in range(n)])
            for dpos, spos in copies:
                assert NB <= dpos < NB + PRIV and spos < NB
                ua[dpos] = ua[spos]
            for d in range(3):
                touched = set()
                for pos in ops[d]:
                    assert all(NB <= q < NB + PRIV for q in pos) and not (touched & set(pos))  # lines of a pass are disjoint
                    touched |= set(pos)
                    ua[pos] = W @ ua[pos]
        else:
            assert cm is None or not cm[bco[b]:bco[b + 1]].any()
        cells = np.arange(bco[b], bco[b + 1])
        nc = len(cells)
        pos = np.zeros((nc, nd), dtype=np.int64)
        for c in range(nc):
            for k in range(n):
                w = ixw[b, :, c * n + k]
                pos[c, n2 * k:n2 * (k + 1)] = np.stack([w & 0xffff, w >> 16], axis=1).reshape(-1)[:n2] // 8
        assert pos.max() < NB + PRIV and not np.any(pos == NB - 1)
        sub = o.Desc(pd.dim, pd.degree, pd.n_dofs, pd.loc2glob[cells], pd.JxW[cells], pd.inv_jac[cells],
                     pd.coefficient[cells], pd.constrained, None, pd.dtype, pd.shape_values, pd.shape_gradients, pd.weights)
        loc = o.cell_apply(sub, ua[pos])  # no masks: the hanging nodes are resolved on the batch array
        acc = np.zeros(NB + PRIV)
        np.add.at(acc, pos.reshape(-1), loc.reshape(-1))
        for d in (2, 1, 0):
            for q in ops[d]:
                acc[q] = W.T @ acc[q]
        for dpos, spos in copies:
            acc[spos] += acc[dpos]
        for q in range(JI):
            assert not con[q] or acc[q] == 0.0  # (a constrained dof in a dst slot is padding: pass 2 rewrites it)
            dst[g[q]] = (dst[g[q]] if add else 0.0) + acc[q]
        halo[hoff[b]:hoff[b] + NB - JI] = acc[JI:NB]
    sd, so, si = plan.sdofs, plan.s_off, plan.s_idx
    for i in range(len(sd)):
        gg = int(sd[i] & 0x7fffffff)
        val = src[gg] if (sd[i] >> 31) else sum(halo[si[so[i]:so[i + 1]]])
        dst[gg] = (dst[gg] if add else 0.0) + val
    for oo in plan.orphans:
        gg = int(oo & 0x7fffffff)
        s = src[gg] if (oo >> 31) else 0.0
        dst[gg] = (dst[gg] if add else 0.0) + s
    return dst
