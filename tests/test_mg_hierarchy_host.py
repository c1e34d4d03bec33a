"""Host-side test of the adaptive multigrid hierarchy (mfgpu_mg_hierarchy_create, csrc/mfgpu_mg_hierarchy.cpp): level
meshes, refinement-edge dofs, transfer arrays over the refined parents and copy_to_mg pairs against an independent
Python construction from the same octree (the one tests/test_gpu_mg_adaptive.py runs its V-cycle on); and the
vertex-balanced variant of the ADAPTIVE_GRID recipe."""
import numpy as np
import pytest

import pymfgpu as mf


def _python_hierarchy(mesh, dim, p):
    cl = mesh.cell_levels().astype(np.int64)
    Lmax = int(cl[:, 0].max())
    S = [set() for _ in range(Lmax + 1)]
    for L, cx, cy, cz in cl:
        for l in range(L + 1):
            S[l].add((l, cx >> (L - l), cy >> (L - l), cz >> (L - l)))
    n = p + 1
    lidx = np.stack(np.meshgrid(*[np.arange(n)] * dim, indexing="ij"), axis=-1)[..., ::-1].reshape(-1, dim)
    out = []
    for l in range(Lmax + 1):
        M = mf.Mesh.from_leaves(dim, p, np.array(sorted(S[l]), dtype=np.uint32))
        cells = [tuple(int(v) for v in r) for r in M.cell_levels()]
        idx = {c: k for k, c in enumerate(cells)}
        l2g = M.arrays()["loc2glob"]
        E = set()
        for c, k in idx.items():
            for d in range(dim):
                for side in (0, 1):
                    nb = list(c)
                    nb[1 + d] += 1 if side else -1
                    if not (0 <= nb[1 + d] < (1 << l)) or tuple(nb) in idx:
                        continue
                    E.update(int(g) for g in l2g[k][lidx[:, d] == (p if side else 0)])
        out.append((M, idx, np.array(sorted(E), dtype=np.uint32)))
    return cl, out


@pytest.mark.parametrize("dim,p,nref", [(2, 2, 4), (2, 3, 5), (3, 2, 4), (3, 1, 5)])
def test_hierarchy_matches_the_python_construction(dim, p, nref):
    mesh = mf.Mesh.adaptive_mg(dim, p, nref)
    H = mf.MgHierarchy(mesh)
    cl, ref = _python_hierarchy(mesh, dim, p)
    assert H.n_levels == len(ref)
    nd, nfd = (p + 1) ** dim, (2 * p + 1) ** dim
    al2g = mesh.arrays()["loc2glob"]
    for l, (M, idx, E) in enumerate(ref):
        L = H.level_mesh(l)
        assert L.n_dofs == M.n_dofs and L.n_cells == M.n_cells
        np.testing.assert_array_equal(L.arrays()["loc2glob"], M.arrays()["loc2glob"])
        np.testing.assert_array_equal(H.edge_dofs(l), E)
        # copy pairs: active cells of this level, off the edge
        pa, pl = H.copy_pairs(l)
        want = []
        for k, (Lc, cx, cy, cz) in enumerate(cl):
            if Lc != l:
                continue
            lev = M.arrays()["loc2glob"][idx[(int(Lc), int(cx), int(cy), int(cz))]]
            keep = ~np.isin(lev, E)
            want += list(zip(al2g[k][keep], lev[keep]))
        assert sorted(zip(pa.tolist(), pl.tolist())) == sorted((int(a), int(b)) for a, b in want)
        if l > 0:
            cd, fd = H.transfer_arrays(l, nd, nfd)
            Mc, idxc, _ = ref[l - 1]
            parents = [k for c, k in sorted(idxc.items(), key=lambda kv: kv[1])
                       if (l, 2 * c[1], 2 * c[2], 2 * c[3] if dim == 3 else 0) in idx]
            np.testing.assert_array_equal(cd, Mc.arrays()["loc2glob"][parents])
            assert len(np.unique(fd)) == M.n_dofs  # every level cell is somebody's child: the patches cover the level
    # every unconstrained active dof is copied on exactly one level
    seen = np.zeros(mesh.n_dofs, int)
    for l in range(H.n_levels):
        seen[np.unique(H.copy_pairs(l)[0])] += 1
    con = np.zeros(mesh.n_dofs, bool)
    con[mesh.arrays()["constrained_dofs"]] = True
    assert np.all(seen[~con] == 1)


def test_vertex_balance_keeps_finer_cells_off_the_level_boundary():
    for dim in (2, 3):
        mesh = mf.Mesh.adaptive_mg(dim, 1, 5)
        H = mf.MgHierarchy(mesh)
        for l in range(1, H.n_levels):
            cd, fd = H.transfer_arrays(l, 2 ** dim, 3 ** dim)
            # the coarse dofs the transfer reads are never refinement-edge dofs of the coarse level
            assert not np.isin(cd, H.edge_dofs(l - 1)).any()
