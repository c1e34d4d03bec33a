"""GPU parity of the SURVEY.md 8(f) rows N1 (inverse diagonal, set_constrained_values) and N2 (GpuVector
BLAS-1 / reductions), through the C-ABI, and the caller they exist for: a Jacobi-preconditioned CG in
the shape of poisson.cu:237-260 built from nothing but those entry points."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

import pymfgpu as mf
from oracle import mf_oracle as o
from util import oracle_desc_from_mesh

pytestmark = pytest.mark.gpu

TOL = {mf.F64: 1e-12, mf.F32: 2e-5}


def dv(a, nt=mf.F64):
    v = mf.DeviceVector(len(a), nt)
    v.from_host(a)
    return v


@pytest.mark.parametrize("nt", [mf.F64, mf.F32])
@pytest.mark.parametrize("dim,p,n", [(2, 1, 7), (2, 2, 16), (2, 4, 5), (2, 6, 3), (3, 1, 5), (3, 2, 5), (3, 3, 3),
                                     (3, 4, 4), (3, 5, 2), (3, 6, 2)])
def test_inverse_diagonal_matches_oracle(dim, p, n, nt):
    mesh = mf.Mesh.uniform(dim, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    op = mf.Operator(mesh.desc, mesh)
    d = mf.DeviceVector(mesh.n_dofs, nt)
    d.fill(123.0)  # must be overwritten (inv_diag.reinit, laplace_operator_gpu.h:407)
    op.compute_inverse_diagonal(d)
    mf.synchronize()
    ref = o.compute_inverse_diagonal(od)
    np.testing.assert_allclose(d.to_host(), ref, rtol=TOL[nt])
    # and against the assembled matrix (independent of the oracle's closed form)
    np.testing.assert_allclose(1.0 / d.to_host().astype(np.float64), o.assemble(od).diagonal(), rtol=10 * TOL[nt])


@pytest.mark.parametrize("dim,p,nref", [(2, 2, 4), (2, 4, 4), (3, 1, 4), (3, 2, 4), (3, 4, 4)])
def test_inverse_diagonal_with_hanging_nodes(dim, p, nref):
    mesh = mf.Mesh.adaptive(dim, p, nref)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    op = mf.Operator(mesh.desc, mesh)
    d = mf.DeviceVector(mesh.n_dofs)
    op.compute_inverse_diagonal(d)
    mf.synchronize()
    got, ref = d.to_host(), o.compute_inverse_diagonal(od)
    fin = np.isfinite(ref)            # dofs no cell touches and that are not constrained would be 1/0
    assert fin.all() and np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, rtol=1e-12)


def test_set_constrained_values():
    mesh = mf.Mesh.uniform(3, 2, 4)
    op = mf.Operator(mesh.desc, mesh)
    x = np.random.default_rng(0).standard_normal(mesh.n_dofs)
    v = dv(x)
    op.set_constrained_values(v, -3.5)
    mf.synchronize()
    c = mesh.arrays()["constrained_dofs"]
    exp = x.copy()
    exp[c] = -3.5
    np.testing.assert_array_equal(v.to_host(), exp)


@pytest.mark.parametrize("nt", [mf.F64, mf.F32])
@pytest.mark.parametrize("n", [1, 63, 64, 257, 1000, 524289, 3000001])
def test_vector_operations_match_numpy(n, nt):
    rng = np.random.default_rng(n)
    t = mf.np_dtype(nt)
    a, b, c = (rng.standard_normal(n).astype(t) for _ in range(3))
    b[np.abs(b) < 0.1] = 0.5  # divisor / inverted vector away from 0
    tol = 1e-13 if nt == mf.F64 else 1e-6
    va, vb, vc = dv(a, nt), dv(b, nt), dv(c, nt)
    va.sadd(0.5, -2.0, vb)
    ref = (t(0.5) * a + t(-2.0) * b).astype(t)
    np.testing.assert_allclose(va.to_host(), ref, rtol=tol, atol=tol)
    va.equ(3.0, vc)
    np.testing.assert_array_equal(va.to_host(), t(3.0) * c)
    va.scale(vb)
    ref = (t(3.0) * c) * b
    np.testing.assert_allclose(va.to_host(), ref, rtol=tol)
    va.divide(vb)
    ref = ref / b
    np.testing.assert_allclose(va.to_host(), ref, rtol=tol)
    vb.invert()
    np.testing.assert_allclose(vb.to_host(), t(1.0) / b, rtol=tol)
    va.mul(-0.25)
    ref = t(-0.25) * ref
    np.testing.assert_allclose(va.to_host(), ref, rtol=tol)
    # reductions: accumulate in double, fixed order -> bitwise repeatable, close to numpy's pairwise sum
    x = va.to_host().astype(np.float64)
    y = vc.to_host().astype(np.float64)
    rtol = 1e-12 if nt == mf.F64 else 1e-6
    d1 = va.dot(vc)
    assert d1 == va.dot(vc)
    assert abs(d1 - x @ y) <= rtol * np.linalg.norm(x) * np.linalg.norm(y)
    assert abs(va.l2_norm() - np.linalg.norm(x)) <= rtol * np.linalg.norm(x)
    r = va.add_and_dot(0.5, vc, vb)       # va += 0.5 vc ; va . vb
    x2 = (va.to_host()).astype(np.float64)
    np.testing.assert_allclose(x2, (x.astype(t) + t(0.5) * c).astype(np.float64), rtol=tol, atol=tol)
    z = vb.to_host().astype(np.float64)
    assert abs(r - x2 @ z) <= rtol * np.linalg.norm(x2) * np.linalg.norm(z)
    assert not va.all_zero()
    va.mul(0.0)
    assert va.all_zero()
    mf.synchronize()


@pytest.mark.parametrize("dim,p,n,adaptive", [(2, 2, 16, False), (3, 2, 6, False), (3, 4, 3, False), (2, 2, 4, True),
                                              (3, 2, 4, True)])
def test_jacobi_pcg_solve_like_poisson_driver(dim, p, n, adaptive):
    """poisson.cu:223-260 in miniature: compute_diagonal, then preconditioned CG on A x = b with the
    vector operations of the boundary only.  The solution is checked against a sparse direct solve of the
    oracle's assembled operator (same semantics: identity rows on constrained dofs)."""
    mesh = mf.Mesh.adaptive(dim, p, n) if adaptive else mf.Mesh.uniform(dim, p, n)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    op = mf.Operator(mesh.desc, mesh)
    N = mesh.n_dofs
    rng = np.random.default_rng(3)
    b_host = rng.standard_normal(N)
    b_host[od.constrained] = 0.0          # homogeneous Dirichlet / hanging rows (constraints.set_zero)
    A = o.assemble(od).tocsc()
    x_ref = spla.spsolve(A, b_host)

    b, x, r, z, pv, q, dinv = (mf.DeviceVector(N) for _ in range(7))
    b.from_host(b_host)
    op.compute_inverse_diagonal(dinv)
    r.equ(1.0, b)                          # x0 = 0
    z.equ(1.0, r)
    z.scale(dinv)                          # DiagonalMatrix::vmult
    pv.equ(1.0, z)
    rz = r.dot(z)
    r0 = r.l2_norm()
    its = 0
    for its in range(1, 2000):
        op.vmult(q, pv)
        alpha = rz / pv.dot(q)
        x.add(alpha, pv)
        r.add(-alpha, q)
        if r.l2_norm() <= 1e-12 * r0:
            break
        z.equ(1.0, r)
        z.scale(dinv)
        rz_new = r.dot(z)
        pv.sadd(rz_new / rz, 1.0, z)
        rz = rz_new
    mf.synchronize()
    assert its < 1999, "CG did not converge"
    got = x.to_host()
    assert np.linalg.norm(got - x_ref) <= 1e-8 * np.linalg.norm(x_ref)


def test_pcg_driver_binaries():
    """C++ shim (reference class names: GpuVector BLAS-1, LaplaceOperatorGpu::compute_diagonal /
    get_diagonal_inverse, DiagonalMatrix) + the pcg driver: converges and reproduces the manufactured
    solution on uniform and adaptive meshes; output line as poisson.cu:271-272."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = os.path.join(root, "dealii-cuda_amd", "host", "bin")
    for exe, args, ndofs in (("pcg-2d-p2", ["5"], 65 ** 2), ("pcg-3d-p4", ["3"], 33 ** 3), ("pcg-3d-p4", ["4", "1"], None)):
        out = subprocess.run([os.path.join(b, exe)] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        f = out.stdout.split()
        assert f[0] == exe[4] and f[1] == exe[-1]
        if ndofs:
            assert int(f[2]) == ndofs
        assert 1 < int(f[3]) < 10000 and float(f[5]) < 1e-8
