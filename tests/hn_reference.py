"""Independent reference for hanging-node meshes (test infrastructure).

Builds the CONFORMING finite element operator on a one-irregular quadtree/octree directly from
geometry: nodes are identified by position, a node lying on the boundary of a leaf cell without being
one of that cell's nodes is hanging and equals the cell's polynomial there (u_h = sum_j phi_j^K(x) u_j),
A = C^T K C on the master nodes, Dirichlet rows/columns -> identity.  Nothing here uses the
constraint masks or the loc2glob substitution of hanging_nodes.cuh, so it checks them end to end."""
import numpy as np
import scipy.sparse as sp

from oracle import mf_oracle as o


def _key(x):
    return tuple(np.round(x * 2.0 ** 30).astype(np.int64))


def conforming_operator(dim, p, leaves, coefficient=o.coefficient_value):
    n = p + 1
    nd = n ** dim
    sv, sg, xq, wq, nodes = o.shape_info(p)
    leaves = np.asarray(leaves, dtype=np.int64).reshape(-1, 4)
    lidx = np.stack(np.meshgrid(*[np.arange(n)] * dim, indexing="ij"), axis=-1)[..., ::-1].reshape(-1, dim)
    ids, coords, cell_nodes, node_level = {}, [], [], []
    for lv, *c in leaves:
        h = 2.0 / 2 ** lv
        cn = []
        for li in lidx:
            x = -1.0 + h * (np.array(c[:dim]) + nodes[li])
            k = _key(x)
            if k not in ids:
                ids[k] = len(coords)
                coords.append(x)
                node_level.append(lv)
            node_level[ids[k]] = min(node_level[ids[k]], lv)  # coarsest cell owning the node
            cn.append(ids[k])
        cell_nodes.append(cn)
    coords = np.array(coords)
    N = len(coords)
    cell_nodes = np.array(cell_nodes)
    # element matrices through the oracle's dense tables
    hs = 2.0 / 2.0 ** leaves[:, 0]
    qp = -1.0 + hs[:, None, None] * (leaves[:, None, 1:1 + dim] + xq[lidx][None, :, :])
    wflat = np.ones(nd)
    for d in range(dim):
        wflat = wflat * wq[lidx[:, d]]
    od = o.Desc(dim, p, N, cell_nodes, wflat[None, :] * hs[:, None] ** dim, 1.0 / hs, coefficient(qp),
                np.zeros(0, dtype=np.uint32))
    K = o.assemble(od)  # no constraints: plain assembled stiffness
    # hanging nodes: on the closed boundary of a leaf that is COARSER than every cell owning the node,
    # and not one of that leaf's nodes
    node_level = np.array(node_level)
    hanging = {}
    for ci, (lv, *c) in enumerate(leaves):
        h = 2.0 / 2 ** lv
        lo = -1.0 + h * np.array(c[:dim])
        xi = (coords - lo) / h  # reference coordinates w.r.t. this cell
        inside = np.all((xi > -1e-12) & (xi < 1 + 1e-12), axis=1)
        onb = inside & np.any((np.abs(xi) < 1e-12) | (np.abs(xi - 1) < 1e-12), axis=1)
        mine = np.zeros(N, bool)
        mine[cell_nodes[ci]] = True
        for g in np.nonzero(onb & ~mine & (node_level > lv))[0]:
            if g in hanging:
                continue
            vals = [o.lagrange_eval(nodes, np.clip(xi[g, d], 0, 1))[0][:, 0] for d in range(dim)]
            w = np.ones(nd)
            for d in range(dim):
                w = w * vals[d][lidx[:, d]]
            keep = np.abs(w) > 1e-14
            hanging[g] = (cell_nodes[ci][keep], w[keep])
    is_h = np.zeros(N, bool)
    is_h[list(hanging)] = True
    chained = any(is_h[m].any() for m, _ in hanging.values())
    rows, cols, vals = [], [], []
    for g in range(N):
        if g in hanging:
            m, w = hanging[g]
            rows += [g] * len(m)
            cols += list(m)
            vals += list(w)
        else:
            rows.append(g)
            cols.append(g)
            vals.append(1.0)
    C = sp.csr_matrix((vals, (rows, cols)), shape=(N, N))
    if chained:  # resolve chains like ConstraintMatrix::close()
        for _ in range(4):
            C = C @ C
    dirichlet = np.any(np.abs(np.abs(coords) - 1.0) < 1e-12, axis=1)
    master = ~is_h
    free = master & ~dirichlet
    D = sp.diags(free.astype(float))
    A = D @ (C.T @ K @ C) @ D + sp.diags((master & dirichlet).astype(float))
    return dict(A=A.tocsr(), coords=coords, master=master, free=free, chained=chained, n_hanging=int(is_h.sum()))


def match_by_position(coords_a, coords_b):
    """index into b for every row of a (positions must exist in b)"""
    kb = {_key(x): i for i, x in enumerate(coords_b)}
    return np.array([kb[_key(x)] for x in coords_a])
