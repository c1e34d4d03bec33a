"""GPU parity of the multigrid level transfer (mg_transfer_matrix_free_gpu.cu:391-660) through the
C-ABI against the oracle's restatement of the reference's cell loops, as the reference's own test_mg_transfer.cc does
against deal.II's CPU transfer: random vectors, every level pair, prolongate and restrict_and_add.  Then the pieces
together: a multigrid V-cycle (Chebyshev-Jacobi smoother, CG on the coarsest level) as CG preconditioner, level
operators = mfgpu handles of the level meshes (laplace_operator_gpu.h:154-186), as poisson_mg.cu assembles it.
Tolerance: relative l2 <= 1e-13 (double), 1e-5 (float)."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from test_gpu import rel
from util import oracle_desc_from_mesh

pytestmark = pytest.mark.gpu


def _run(t, n_fine, n_coarse, xc, yf, z, nt=mf.F64):
    a, b = mf.DeviceVector(n_coarse, nt), mf.DeviceVector(n_fine, nt)
    a.from_host(xc)
    b.fill(3.0)  # prolongate overwrites
    t.prolongate(b, a)
    mf.synchronize()
    pro = b.to_host()
    b.from_host(yf)
    a.from_host(z)
    t.restrict_and_add(a, b)
    mf.synchronize()
    return pro, a.to_host()


@pytest.mark.parametrize("nt,tol", [(mf.F64, 1e-13), (mf.F32, 1e-5)])
@pytest.mark.parametrize("dim,p,n", [(2, 1, 8), (2, 2, 5), (2, 3, 4), (2, 4, 6), (2, 6, 2), (3, 1, 4), (3, 2, 3), (3, 3, 2),
                                     (3, 4, 3), (3, 5, 1), (3, 6, 1)])
def test_transfer_matches_oracle_on_cubes(dim, p, n, nt, tol):
    mc, mfine = mf.Mesh.uniform(dim, p, n, number_type=nt), mf.Mesh.uniform(dim, p, 2 * n, number_type=nt)
    C, F = o.uniform_mesh_desc(dim, p, n), o.uniform_mesh_desc(dim, p, 2 * n)
    cd, fd = o.mg_patches_uniform(dim, p, n, C, F)
    rng = np.random.default_rng(dim * 100 + p * 10 + n)
    dt = mf.np_dtype(nt)
    x, y, z = (rng.random(k).astype(dt).astype(np.float64) for k in (C.n_dofs, F.n_dofs, C.n_dofs))
    want_p = o.mg_prolongate(dim, p, cd, fd, F.n_dofs, C.constrained, x)
    want_r = o.mg_restrict_and_add(dim, p, cd, fd, C.n_dofs, C.constrained, z, y)
    for t in (mf.Transfer.from_meshes(mc, mfine),
              mf.Transfer.from_arrays(dim, p, cd, fd, C.n_dofs, F.n_dofs, C.constrained, nt, o.mg_prolongation_1d(p))):
        got_p, got_r = _run(t, F.n_dofs, C.n_dofs, x, y, z, nt)
        assert rel(got_p, want_p) <= tol
        assert rel(got_r, want_r) <= tol
        np.testing.assert_array_equal(got_r[C.constrained], z.astype(dt)[C.constrained])
        assert t.memory_consumption() > 0


@pytest.mark.parametrize("dim,p,r", [(2, 2, 2), (2, 4, 1), (3, 1, 1), (3, 2, 1), (3, 4, 0), (3, 4, 1)])
def test_transfer_on_the_ball(dim, p, r):
    mc, mfine = mf.Mesh.ball(dim, p, r), mf.Mesh.ball(dim, p, r + 1)
    cd, fd = mc.transfer_patches(mfine)
    con = mc.arrays()["constrained_dofs"].copy()
    rng = np.random.default_rng(5)
    x, y, z = rng.random(mc.n_dofs), rng.random(mfine.n_dofs), rng.random(mc.n_dofs)
    got_p, got_r = _run(mf.Transfer.from_meshes(mc, mfine), mfine.n_dofs, mc.n_dofs, x, y, z)
    assert rel(got_p, o.mg_prolongate(dim, p, cd, fd, mfine.n_dofs, con, x)) <= 1e-13
    assert rel(got_r, o.mg_restrict_and_add(dim, p, cd, fd, mc.n_dofs, con, z, y)) <= 1e-13


def test_transfer_errors_are_loud():
    mc, mfine = mf.Mesh.uniform(2, 2, 3), mf.Mesh.uniform(2, 2, 7)
    with pytest.raises(mf.MfgpuError):
        mf.Transfer.from_meshes(mc, mfine)
    with pytest.raises(mf.MfgpuError):
        mf.Transfer.from_arrays(2, 2, np.zeros((1, 9), np.uint32), np.full((1, 25), 99, np.uint32), 9, 25, [])


class _Level:
    def __init__(self, mesh):
        self.mesh, self.n = mesh, mesh.n_dofs
        self.op = mf.Operator(mesh.desc, mesh)
        self.dinv = mf.DeviceVector(self.n)
        self.op.compute_inverse_diagonal(self.dinv)
        self.r, self.x, self.t, self.d = (mf.DeviceVector(self.n) for _ in range(4))
        # largest eigenvalue of D^-1 A by power iteration (deal.II estimates it with 15 CG steps, poisson_mg.cu:353-356)
        v, w = mf.DeviceVector(self.n), mf.DeviceVector(self.n)
        v.from_host(np.random.default_rng(0).standard_normal(self.n))
        lam = 1.0
        for _ in range(20):
            self.op.vmult(w, v)
            w.scale(self.dinv)
            lam = w.l2_norm() / v.l2_norm()
            v.equ(1.0 / w.l2_norm(), w)
        self.lmax = 1.2 * lam
        self.lmin = self.lmax / 15.0  # smoothing_range = 15

    def chebyshev(self, x, b, degree=5, zero_start=False):
        """PreconditionChebyshev (degree 5) with the inverse diagonal as inner preconditioner"""
        theta, delta = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        r, d, t = self.r, self.d, self.t
        if zero_start:
            r.equ(1.0, b)
        else:
            self.op.vmult(t, x)
            r.equ(1.0, b)
            r.add(-1.0, t)
        d.equ(1.0 / theta, r)
        d.scale(self.dinv)
        if zero_start:
            x.equ(1.0, d)
        else:
            x.add(1.0, d)
        for _ in range(degree - 1):
            self.op.vmult(t, d)
            r.add(-1.0, t)
            rho_new = 1.0 / (2.0 * sigma - rho)
            t.equ(2.0 * rho_new / delta, r)
            t.scale(self.dinv)
            d.sadd(rho_new * rho, 1.0, t)
            x.add(1.0, d)
            rho = rho_new


def _vcycle(levels, transfers, l, x, b):
    L = levels[l]
    if l == 0:  # MGCoarseIterative: unpreconditioned CG (poisson_mg.cu:61-83)
        x.fill(0.0)
        r, p_, q = L.r, L.d, L.t
        r.equ(1.0, b)
        p_.equ(1.0, r)
        rr, r0 = r.dot(r), r.l2_norm()
        for _ in range(2000):
            if r0 == 0.0 or np.sqrt(rr) <= 1e-10 * r0:
                break
            L.op.vmult(q, p_)
            alpha = rr / p_.dot(q)
            x.add(alpha, p_)
            r.add(-alpha, q)
            rr_new = r.dot(r)
            p_.sadd(rr_new / rr, 1.0, r)
            rr = rr_new
        return
    L.chebyshev(x, b, zero_start=True)                   # pre-smoothing
    res, C = mf.DeviceVector(L.n), levels[l - 1]
    L.op.vmult(res, x)
    res.sadd(-1.0, 1.0, b)                               # residual
    bc, xc = mf.DeviceVector(C.n), mf.DeviceVector(C.n)  # zero-filled
    transfers[l - 1].restrict_and_add(bc, res)
    _vcycle(levels, transfers, l - 1, xc, bc)
    transfers[l - 1].prolongate(res, xc)
    x.add(1.0, res)                                      # coarse-grid correction
    L.chebyshev(x, b)                                    # post-smoothing


@pytest.mark.parametrize("dim,p,n0,nlev", [(2, 2, 2, 4), (3, 2, 1, 4), (3, 4, 1, 4)])
def test_multigrid_preconditioned_cg(dim, p, n0, nlev):
    """CG + one V-cycle per iteration solves the variable-coefficient Poisson problem of poisson.cu to 1e-10 in a
    number of iterations that does not grow with the level count, and to the sparse direct solution"""
    import scipy.sparse.linalg as spla

    meshes = [mf.Mesh.uniform(dim, p, n0 * 2 ** l) for l in range(nlev)]
    iters = []
    for top in (nlev - 1, nlev):
        levels = [_Level(m) for m in meshes[:top]]
        transfers = [mf.Transfer.from_meshes(meshes[l], meshes[l + 1]) for l in range(top - 1)]
        F = levels[-1]
        od = oracle_desc_from_mesh(F.mesh)
        bh = np.random.default_rng(1).standard_normal(F.n)
        bh[od.constrained] = 0.0
        b, x, r, z, pv, q = (mf.DeviceVector(F.n) for _ in range(6))
        b.from_host(bh)
        r.equ(1.0, b)
        _vcycle(levels, transfers, top - 1, z, r)
        pv.equ(1.0, z)
        rz, r0 = r.dot(z), r.l2_norm()
        for it in range(1, 60):
            F.op.vmult(q, pv)
            alpha = rz / pv.dot(q)
            x.add(alpha, pv)
            r.add(-alpha, q)
            if r.l2_norm() <= 1e-10 * r0:
                break
            _vcycle(levels, transfers, top - 1, z, r)
            rz_new = r.dot(z)
            pv.sadd(rz_new / rz, 1.0, z)
            rz = rz_new
        iters.append(it)
        if F.n <= 40000:
            A = o.assemble(od).tocsc()
            xr = spla.spsolve(A, bh)
            assert np.linalg.norm(x.to_host() - xr) <= 1e-8 * np.linalg.norm(xr)
    assert iters[-1] <= 14 and iters[-1] <= iters[0] + 2, iters


def test_poisson_mg_driver_binaries():
    """C++ shim (mfgpu_shim_mg.h) + poisson_mg driver: CG + V-cycle to 1e-12, checked against the known solution;
    line: dim degree n_dofs levels cg_iterations wall rel_error"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = os.path.join(root, "dealii-cuda_amd", "host", "bin")
    its = {}
    for exe, arg, ndofs in (("poisson-mg-2d-p2", "5", 65 ** 2), ("poisson-mg-2d-p2", "7", 257 ** 2),
                            ("poisson-mg-3d-p4", "3", 33 ** 3), ("poisson-mg-3d-p4", "4", 65 ** 3),
                            ("poisson-mg-3d-p2-ball", "3", None)):
        out = subprocess.run([os.path.join(b, exe), arg], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        f = out.stdout.split()
        if ndofs is not None:
            assert int(f[2]) == ndofs
        assert int(f[3]) == int(arg) + 1 and float(f[6]) < 1e-8
        its[(exe, arg)] = int(f[4])
    # ADAPTIVE_GRID: local smoothing with refinement-edge matrices on the pseudo-adaptive mesh (vertex-balanced), the
    # active-mesh operator with hanging nodes as system matrix
    for exe, args in (("poisson-mg-2d-p2-adaptive", ("4", "6")), ("poisson-mg-3d-p4-adaptive", ("4", "5"))):
        got = []
        for arg in args:
            out = subprocess.run([os.path.join(b, exe), arg], capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stdout + out.stderr
            f = out.stdout.split()
            assert float(f[6]) < 1e-8 and int(f[3]) >= 5
            got.append(int(f[4]))
        assert got[1] <= got[0] + 3 <= 18, got
    # level-independent iteration counts
    assert its[("poisson-mg-2d-p2", "7")] <= its[("poisson-mg-2d-p2", "5")] + 2 <= 16
    assert its[("poisson-mg-3d-p4", "4")] <= its[("poisson-mg-3d-p4", "3")] + 2 <= 16
    assert its[("poisson-mg-3d-p2-ball", "3")] <= 30
