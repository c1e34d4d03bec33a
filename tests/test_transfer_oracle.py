"""Host-side tests of the multigrid level transfer.  The oracle restates
MGTransferMatrixFreeGpu::prolongate / restrict_and_add (mg_transfer_matrix_free_gpu.cu:595-660) in the reference's own
cell-loop-with-weights form; it is pinned by what the reference's own test checks against deal.II's CPU transfer
(test_mg_transfer.cc) can be replaced with here: the embedding property -- a polynomial of degree <= p per direction
prolongates exactly to its nodal values on the fine level (independent of all index logic: only dof coordinates) --
and <P x, y> = <x, P^T y>.  The mesh stand-in's patch arrays are checked against the oracle's independent builder."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o


@pytest.mark.parametrize("dim,p,n", [(2, 1, 3), (2, 2, 4), (2, 4, 2), (3, 1, 2), (3, 2, 2), (3, 3, 1), (3, 4, 2)])
def test_oracle_transfer_embedding_and_adjointness(dim, p, n):
    C, F = o.uniform_mesh_desc(dim, p, n), o.uniform_mesh_desc(dim, p, 2 * n)
    cd, fd = o.mg_patches_uniform(dim, p, n, C, F)

    def poly(x):
        return np.prod([(0.7 + 0.3 * (d + 1) * x[..., d]) ** p for d in range(dim)], axis=0)

    uf = o.mg_prolongate(dim, p, cd, fd, F.n_dofs, [], poly(C.dof_coords))
    assert np.abs(uf - poly(F.dof_coords)).max() <= 1e-13 * np.abs(uf).max()
    rng = np.random.default_rng(p)
    x, y, z = rng.standard_normal(C.n_dofs), rng.standard_normal(F.n_dofs), rng.standard_normal(C.n_dofs)
    Px = o.mg_prolongate(dim, p, cd, fd, F.n_dofs, C.constrained, x)
    Rty = o.mg_restrict_and_add(dim, p, cd, fd, C.n_dofs, C.constrained, z, y)
    # Dirichlet semantics: coarse boundary dofs are read as 0 by prolongate and left alone by restrict_and_add
    x0 = x.copy()
    x0[C.constrained] = 0.0
    np.testing.assert_allclose(Px, o.mg_prolongate(dim, p, cd, fd, F.n_dofs, [], x0), atol=1e-14)
    np.testing.assert_array_equal(Rty[C.constrained], z[C.constrained])
    assert abs(y @ Px - x0 @ (Rty - z)) <= 1e-12 * np.linalg.norm(y) * np.linalg.norm(Px)
    # 1D matrix: rows sum to one (partition of unity), coincident nodes are copied
    P1 = o.mg_prolongation_1d(p)
    np.testing.assert_allclose(P1.sum(axis=1), 1.0, atol=1e-14)
    assert abs(P1[0, 0] - 1) < 1e-14 and abs(P1[2 * p, p] - 1) < 1e-14 and abs(P1[p, :].sum() - 1) < 1e-14


@pytest.mark.parametrize("dim,p,n", [(2, 2, 3), (2, 4, 2), (3, 1, 3), (3, 4, 2), (3, 6, 1)])
def test_mesh_standin_patches_match_the_oracle_builder(dim, p, n):
    mc, mfine = mf.Mesh.uniform(dim, p, n), mf.Mesh.uniform(dim, p, 2 * n)
    cd, fd = mc.transfer_patches(mfine)
    C, F = o.uniform_mesh_desc(dim, p, n), o.uniform_mesh_desc(dim, p, 2 * n)
    ocd, ofd = o.mg_patches_uniform(dim, p, n, C, F)
    np.testing.assert_array_equal(cd, ocd)
    np.testing.assert_array_equal(fd, ofd)
    with pytest.raises(mf.MfgpuError):
        mc.transfer_patches(mf.Mesh.uniform(dim, p, 2 * n + 1))
    with pytest.raises(mf.MfgpuError):
        mf.Mesh.adaptive(dim, p, 3).transfer_patches(mfine)


@pytest.mark.parametrize("dim,p,r", [(2, 2, 1), (2, 3, 2), (3, 2, 0), (3, 4, 1)])
def test_ball_patches_are_geometrically_consistent(dim, p, r):
    """ball level pair: every patch entry that coincides with a coarse node (even patch index in every direction for
    equidistant p = 1, 2 nodes; the corners in general) carries the same point as the coarse dof, except on the curved
    boundary where refinement moves new vertices onto the sphere"""
    mc, mfine = mf.Mesh.ball(dim, p, r), mf.Mesh.ball(dim, p, r + 1)
    cd, fd = mc.transfer_patches(mfine)
    nf = 2 * p + 1
    xc, xf = mc.dof_coords(), mfine.dof_coords()
    corners_f = [sum((2 * p if (k >> d) & 1 else 0) * nf ** d for d in range(dim)) for k in range(2 ** dim)]
    corners_c = [sum((p if (k >> d) & 1 else 0) * (p + 1) ** d for d in range(dim)) for k in range(2 ** dim)]
    for kf, kc in zip(corners_f, corners_c):
        np.testing.assert_allclose(xf[fd[:, kf]], xc[cd[:, kc]], atol=1e-13)  # coarse vertices are fine vertices
    assert len(np.unique(fd)) == mfine.n_dofs  # the patches cover the fine level
