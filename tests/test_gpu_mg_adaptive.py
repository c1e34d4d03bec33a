"""Local-smoothing multigrid on the ADAPTIVE stand-in mesh, assembled from the pieces of SURVEY.md 8(f) N4 through the
C-ABI: level meshes (all octree cells of a level), level operators with refinement-edge dofs and their interface
matrices (mfgpu_level_*, laplace_operator_gpu.h:154-186, 306-352), level transfers over the refined parents
(mfgpu_transfer_*, mg_transfer_matrix_free_gpu.cu:391-660), copy_to_mg / copy_from_mg index pairs, Chebyshev smoothers,
and the active-mesh operator with hanging nodes as system matrix -- the configuration poisson_mg.cu builds with deal.II's
Multigrid + PreconditionMG (:199-380).

The V-cycle schedule (where the edge matrices enter) is deal.II library code that is not in the reference tree; it is
restated here from its published algorithm (Janssen & Kanschat, local smoothing with edge matrices) and checked by what
it must deliver: CG preconditioned with it converges to the sparse direct solution in a number of iterations that stays
bounded as the mesh is refined.  Parity of every PIECE with the reference's code is tested elsewhere
(test_gpu_level.py, test_gpu_transfer.py, test_gpu.py)."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from util import oracle_desc_from_mesh

pytestmark = pytest.mark.gpu


class _Smoother:
    def __init__(self, lev, n):
        self.lev, self.n = lev, n
        self.dinv = mf.DeviceVector(n)
        lev.compute_inverse_diagonal(self.dinv)
        self.r, self.t, self.d = (mf.DeviceVector(n) for _ in range(3))
        v, w = mf.DeviceVector(n), mf.DeviceVector(n)
        v.from_host(np.random.default_rng(0).standard_normal(n))
        lam = 1.0
        for _ in range(20):
            lev.vmult(w, v)
            w.scale(self.dinv)
            lam = w.l2_norm() / v.l2_norm()
            v.equ(1.0 / w.l2_norm(), w)
        self.lmax, self.lmin = 1.2 * lam, 1.2 * lam / 15.0

    def smooth(self, x, b, zero_start, degree=5):
        theta, delta = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        r, d, t = self.r, self.d, self.t
        r.equ(1.0, b)
        if not zero_start:
            self.lev.vmult(t, x)
            r.add(-1.0, t)
        d.equ(1.0 / theta, r)
        d.scale(self.dinv)
        if zero_start:
            x.equ(1.0, d)
        else:
            x.add(1.0, d)
        for _ in range(degree - 1):
            self.lev.vmult(t, d)
            r.add(-1.0, t)
            rho_new = 1.0 / (2.0 * sigma - rho)
            t.equ(2.0 * rho_new / delta, r)
            t.scale(self.dinv)
            d.sadd(rho_new * rho, 1.0, t)
            x.add(1.0, d)
            rho = rho_new


def _vertex_balanced_leaves(dim, nref):
    """octree leaves on the cube, refined towards a ball, 2:1 balanced over faces, edges AND vertices
    (Triangulation::limit_level_difference_at_vertices, which the reference's multigrid programs set, poisson_mg.cu:131):
    a cell of level l + 1 then never touches the boundary of the level-l region"""
    import itertools

    leaves = {(2,) + c + (0,) * (3 - dim) for c in itertools.product(range(4), repeat=dim)}

    def refine(c):
        leaves.discard(c)
        for ch in itertools.product((0, 1), repeat=dim):
            leaves.add((c[0] + 1,) + tuple(2 * c[1 + d] + ch[d] for d in range(dim)) + (0,) * (3 - dim))

    def covering(l, pos):
        for lv in range(l, -1, -1):
            c = (lv,) + tuple(pos[d] >> (l - lv) for d in range(dim)) + (0,) * (3 - dim)
            if c in leaves:
                return c
        return None

    for step in range(nref):
        R = 0.9 - 0.2 * step
        for c in [c for c in leaves if c[0] == 2 + step]:
            h = 2.0 / (1 << c[0])
            ctr = [-1.0 + h * (c[1 + d] + 0.5) for d in range(dim)]
            if sum(v * v for v in ctr) < R * R:
                refine(c)
        changed = True
        while changed:
            changed = False
            for c in sorted(leaves, key=lambda c: -c[0]):
                if c not in leaves:
                    continue
                for off in itertools.product((-1, 0, 1), repeat=dim):
                    pos = tuple(c[1 + d] + off[d] for d in range(dim))
                    if any(v < 0 or v >= (1 << c[0]) for v in pos):
                        continue
                    nb = covering(c[0], pos)
                    if nb is not None and nb[0] < c[0] - 1:
                        refine(nb)
                        changed = True
    return np.array(sorted(leaves), dtype=np.uint32)


def _hierarchy(dim, p, nref):
    mesh = mf.Mesh.from_leaves(dim, p, _vertex_balanced_leaves(dim, nref))
    cl = mesh.cell_levels().astype(np.int64)
    Lmax = int(cl[:, 0].max())
    S = [set() for _ in range(Lmax + 1)]
    for L, cx, cy, cz in cl:
        for l in range(L + 1):
            S[l].add((l, cx >> (L - l), cy >> (L - l), cz >> (L - l)))
    n = p + 1
    nd = n ** dim
    lidx = np.stack(np.meshgrid(*[np.arange(n)] * dim, indexing="ij"), axis=-1)[..., ::-1].reshape(-1, dim)  # x fastest
    meshes, index, edges = [], [], []
    for l in range(Lmax + 1):
        leaves = np.array(sorted(S[l]), dtype=np.uint32)
        M = mf.Mesh.from_leaves(dim, p, leaves)
        a = M.arrays()
        assert a["constraint_mask"] is None or not a["constraint_mask"].any()
        cells = [tuple(int(v) for v in r) for r in M.cell_levels()]
        idx = {c: k for k, c in enumerate(cells)}
        # refinement edge of the level: faces whose same-level neighbour is inside the domain but not a cell of the level
        l2g = a["loc2glob"]
        E = set()
        for c, k in idx.items():
            for d in range(dim):
                for side in (0, 1):
                    nb = list(c)
                    nb[1 + d] += 1 if side else -1
                    if not (0 <= nb[1 + d] < (1 << l)) or tuple(nb) in idx:
                        continue
                    E.update(int(g) for g in l2g[k][lidx[:, d] == (p if side else 0)])
        meshes.append(M)
        index.append(idx)
        edges.append(np.array(sorted(E), dtype=np.uint32))
    return mesh, cl, meshes, index, edges, lidx, nd


@pytest.mark.parametrize("dim,p,nrefs", [(2, 2, (2, 3)), (2, 4, (2, 3)), (3, 2, (1, 2))])
def test_local_smoothing_multigrid_on_the_adaptive_mesh(dim, p, nrefs):
    import scipy.sparse.linalg as spla

    iters = []
    for nref in nrefs:
        mesh, cl, meshes, index, edges, lidx, nd = _hierarchy(dim, p, nref)
        nl = len(meshes)
        assert any(len(e) for e in edges)  # there are refinement edges
        levs = [mf.Level(M.desc, E, M) for M, E in zip(meshes, edges)]
        smo = [_Smoother(L, M.n_dofs) for L, M in zip(levs, meshes)]
        nf = 2 * p + 1
        X = np.stack(np.meshgrid(*[np.arange(nf)] * dim, indexing="ij"), axis=-1)[..., ::-1].reshape(-1, dim)
        tr = []
        for l in range(1, nl):
            Mc, Mf = meshes[l - 1], meshes[l]
            l2c, l2f = Mc.arrays()["loc2glob"], Mf.arrays()["loc2glob"]
            cd, fd = [], []
            for c, k in index[l - 1].items():
                kids = [(l,) + tuple(2 * c[1 + d] + ((ch >> d) & 1) if d < dim else 0 for d in range(3)) for ch in range(2 ** dim)]
                if kids[0] not in index[l]:
                    continue
                a_ = (X > p).astype(np.int64)
                child = sum(a_[:, d] << d for d in range(dim))
                local = sum((X[:, d] - a_[:, d] * p) * (p + 1) ** d for d in range(dim))
                kid_idx = np.array([index[l][kk] for kk in kids])
                cd.append(l2c[k])
                fd.append(l2f[kid_idx[child], local])
            tr.append(mf.Transfer.from_arrays(dim, p, np.array(cd), np.array(fd), Mc.n_dofs, Mf.n_dofs,
                                              Mc.arrays()["constrained_dofs"]))
        # copy_to_mg / copy_from_mg pairs: dofs of active cells on their own level, off the level's refinement edge
        al2g = mesh.arrays()["loc2glob"]
        pairs = [([], []) for _ in range(nl)]
        for k, (L, cx, cy, cz) in enumerate(cl):
            kk = index[L][(int(L), int(cx), int(cy), int(cz))]
            lev_dofs = meshes[L].arrays()["loc2glob"][kk]
            keep = ~np.isin(lev_dofs, edges[L])
            pairs[L][0].extend(al2g[k][keep])
            pairs[L][1].extend(lev_dofs[keep])
        od = oracle_desc_from_mesh(mesh)
        A = mf.Operator(mesh.desc, mesh)
        N = mesh.n_dofs
        defect = [mf.DeviceVector(M.n_dofs) for M in meshes]
        defect2 = [mf.DeviceVector(M.n_dofs) for M in meshes]
        sol = [mf.DeviceVector(M.n_dofs) for M in meshes]
        tv = [mf.DeviceVector(M.n_dofs) for M in meshes]

        def coarse(x, b, L):  # CG on level 0
            x.fill(0.0)
            r, pv, q = mf.DeviceVector(L.n), mf.DeviceVector(L.n), mf.DeviceVector(L.n)
            r.equ(1.0, b)
            pv.equ(1.0, r)
            rr = r.dot(r)
            r0 = np.sqrt(rr)
            for _ in range(500):
                if r0 == 0 or np.sqrt(rr) <= 1e-12 * r0:
                    break
                L.lev.vmult(q, pv)
                alpha = rr / pv.dot(q)
                x.add(alpha, pv)
                r.add(-alpha, q)
                rn = r.dot(r)
                pv.sadd(rn / rr, 1.0, r)
                rr = rn

        def v_step(l):
            if l == 0:
                coarse(sol[0], defect[0], smo[0])
                return
            smo[l].smooth(sol[l], defect[l], True)
            levs[l].vmult(tv[l], sol[l])                       # t = A x
            defect2[l].fill(0.0)
            levs[l].vmult_interface_down(defect2[l], sol[l])   # + the rows of the edge dofs (edge_out)
            tv[l].add(1.0, defect2[l])
            tv[l].sadd(-1.0, 1.0, defect[l])                   # t = defect - t
            tr[l - 1].restrict_and_add(defect[l - 1], tv[l])
            sol[l - 1].fill(0.0)
            v_step(l - 1)
            tr[l - 1].prolongate(tv[l], sol[l - 1])
            sol[l].add(1.0, tv[l])
            levs[l].vmult_interface_up(tv[l], sol[l])          # the edge values' action on the interior rows (edge_in)
            defect2[l].equ(1.0, defect[l])
            defect2[l].add(-1.0, tv[l])
            smo[l].smooth(sol[l], defect2[l], False)

        def precondition(z, r):
            rh = r.to_host()
            for l in range(nl):
                d = np.zeros(meshes[l].n_dofs)
                if pairs[l][0]:
                    d[np.array(pairs[l][1])] = rh[np.array(pairs[l][0])]
                defect[l].from_host(d)
            v_step(nl - 1)
            zh = np.zeros(N)
            for l in range(nl):
                if pairs[l][0]:
                    zh[np.array(pairs[l][0])] = sol[l].to_host()[np.array(pairs[l][1])]
            z.from_host(zh)

        bh = np.random.default_rng(1).standard_normal(N)
        bh[od.constrained] = 0.0
        b, x, r, z, pv, q = (mf.DeviceVector(N) for _ in range(6))
        b.from_host(bh)
        r.equ(1.0, b)
        precondition(z, r)
        pv.equ(1.0, z)
        rz, r0 = r.dot(z), r.l2_norm()
        it = 0
        for it in range(1, 80):
            A.vmult(q, pv)
            alpha = rz / pv.dot(q)
            x.add(alpha, pv)
            r.add(-alpha, q)
            if r.l2_norm() <= 1e-10 * r0:
                break
            precondition(z, r)
            rz_new = r.dot(z)
            pv.sadd(rz_new / rz, 1.0, z)
            rz = rz_new
        iters.append(it)
        xr = spla.spsolve(o.assemble(od).tocsc(), bh)
        assert np.linalg.norm(x.to_host() - xr) <= 1e-7 * np.linalg.norm(xr), iters
    print(f"local smoothing MG dim={dim} p={p}: CG iterations {iters}")
    assert iters[-1] <= 25 and iters[-1] <= iters[0] + 5, iters
