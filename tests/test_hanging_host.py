"""CPU: the adaptive mesh stand-in (masks, loc2glob substitution) + the oracle's emulation of the GPU
path against an independent conforming-space operator built from geometry (tests/hn_reference.py)."""
import numpy as np
import pytest

import pymfgpu as mf
from hn_reference import conforming_operator, match_by_position
from oracle import mf_oracle as o
from util import oracle_desc_from_mesh


def leaves_refine(dim, base_level, refine_cells):
    """uniform level `base_level`, then refine the listed cells (given as coords at base level)"""
    n = 2 ** base_level
    out = []
    rs = {tuple(c) for c in refine_cells}
    rng = [range(n)] * dim + [range(1)] * (3 - dim)
    for cz in rng[2]:
        for cy in rng[1]:
            for cx in rng[0]:
                c = (cx, cy, cz)[:dim]
                if c in rs:
                    for k in range(2 ** dim):
                        ch = [2 * c[d] + ((k >> d) & 1) for d in range(dim)] + [0] * (3 - dim)
                        out.append([base_level + 1] + ch)
                else:
                    out.append([base_level] + list(c) + [0] * (3 - dim))
    return out


CASES = {
    # 2D: one refined cell in a 2x2 / 4x4 grid, an L-shaped refined region
    "2d_corner": (2, leaves_refine(2, 1, [(0, 0)])),
    "2d_center": (2, leaves_refine(2, 2, [(1, 1), (2, 1)])),
    "2d_L": (2, leaves_refine(2, 2, [(0, 0), (1, 0), (0, 1), (3, 3)])),
    # 3D: the situations of test_hanging_nodes_gpu.cu:297-331 -- a refined corner cell; three of the four
    # cell columns around the central z-edge refined (the fine cells at that edge have same-level face
    # neighbours but a coarser DIAGONAL neighbour: edge-only constraints); a refined slab
    "3d_corner": (3, leaves_refine(3, 1, [(0, 0, 0)])),
    "3d_edge": (3, leaves_refine(3, 1, [(0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 1, 0), (1, 1, 1)])),
    "3d_slab": (3, leaves_refine(3, 1, [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0)])),
    "3d_two": (3, leaves_refine(3, 2, [(1, 1, 1), (2, 2, 2), (1, 2, 1)])),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_masks_and_substitution_reproduce_conforming_operator(name, p):
    dim, leaves = CASES[name]
    if dim == 3 and p > 2 and name == "3d_two":
        pytest.skip("large dense reference")
    mesh = mf.Mesh.from_leaves(dim, p, leaves)
    a = mesh.arrays()
    assert a["constraint_mask"] is not None and (a["constraint_mask"] != 0).any()
    od = oracle_desc_from_mesh(mesh)
    ref = conforming_operator(dim, p, leaves)
    assert not ref["chained"]
    if name in ("3d_edge",):
        assert (a["constraint_mask"] & (0b111 << 6)).any()       # edge-only constraints are exercised
    # every hanging dof is referenced by no cell after the substitution (orphan, identity row)
    con = np.zeros(mesh.n_dofs, bool)
    con[a["constrained_dofs"]] = True
    touched = np.zeros(mesh.n_dofs, bool)
    touched[a["loc2glob"].reshape(-1)] = True
    xyz = mesh.dof_coords()
    dirichlet = np.any(np.abs(np.abs(xyz) - 1) < 1e-12, axis=1)
    assert not (touched & con & ~dirichlet).any()
    assert (~touched == (con & ~dirichlet) | (~touched & dirichlet)).all()
    # same operator on the free dofs, for a smooth and a random input
    free_p = ~con
    idx = match_by_position(xyz[free_p], ref["coords"])
    assert ref["free"][idx].all() and ref["free"].sum() == free_p.sum()
    rng = np.random.default_rng(3)
    for trial in range(2):
        xm = np.zeros(len(ref["coords"]))
        vals = np.sin(2 * xyz[free_p, 0]) + xyz[free_p, 1] ** 2 if trial == 0 else rng.standard_normal(free_p.sum())
        xm[idx] = vals
        x = np.zeros(mesh.n_dofs)
        x[free_p] = vals
        x[con] = rng.standard_normal(con.sum())      # constrained entries must not influence free rows
        y = o.vmult(od, x)
        ym = ref["A"] @ xm
        assert np.linalg.norm(y[free_p] - ym[idx]) <= 1e-11 * np.linalg.norm(ym)
        np.testing.assert_array_equal(y[con], x[con])  # identity rows, Dirichlet AND hanging


@pytest.mark.parametrize("dim,p,nref", [(2, 2, 3), (2, 3, 4), (3, 1, 4), (3, 2, 4)])
def test_bmop_adaptive_recipe_is_one_irregular_and_symmetric(dim, p, nref):
    """pseudo_adaptive_refinement (bmop_common.h:49-105): level jumps <= 1 over faces (and edges in
    3D), operator symmetric positive semi-definite on the free dofs"""
    mesh = mf.Mesh.adaptive(dim, p, nref)
    lv = mesh.cell_levels()
    assert len(np.unique(lv[:, 0])) >= 2
    od = oracle_desc_from_mesh(mesh)
    con = np.zeros(mesh.n_dofs, bool)
    con[od.constrained] = True
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    u[con] = 0
    v[con] = 0
    Au, Av = o.vmult(od, u), o.vmult(od, v)
    assert abs(v @ Au - u @ Av) <= 1e-11 * abs(v @ Au)
    assert u @ Au > 0
    # constants are in the kernel on rows away from the Dirichlet boundary -- THROUGH hanging faces too
    y = o.vmult(od, np.where(con, 0.0, 1.0) + np.where(con, 1.0, 0.0) * 0)
    ones = np.ones(mesh.n_dofs)
    ones_in = ones.copy()
    touches_dirichlet = np.zeros(mesh.n_dofs, bool)
    xyz = mesh.dof_coords()
    dirichlet = np.any(np.abs(np.abs(xyz) - 1) < 1e-12, axis=1)
    # a constant extended to hanging dofs: hanging values are interpolated from masters = 1, so only
    # Dirichlet zeroing breaks the constant: rows of cells touching no Dirichlet dof must vanish
    for c in range(mesh.n_cells):
        if dirichlet[od.loc2glob[c]].any():
            touches_dirichlet[od.loc2glob[c]] = True
    y = o.vmult(od, ones_in)
    sel = ~touches_dirichlet & ~con
    assert sel.any() and np.abs(y[sel]).max() <= 1e-10 * np.abs(y).max()
