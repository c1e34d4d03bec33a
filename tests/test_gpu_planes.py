"""GPU parity tests of the plane-per-thread cell loop (apply_planes3, mfgpu_kernels_p.hip) against the oracle, through
the C-ABI: batch sizes from one cell to a full wave (64 / n cells), ragged last batches, every supported degree, both
number types, vmult and vmult_add, Dirichlet rows owned by one batch and shared by several, and the kernel families
it replaced on the same inputs (mfgpu_desc.kernel).

Tolerance: relative l2 <= 1e-12 in double, <= 1e-5 in float (north_star / BASELINE.md section 2)."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from test_gpu import TOL, gpu_vmult, rel
from util import desc_from_oracle, oracle_desc_from_mesh

pytestmark = pytest.mark.gpu

# the plane plan's two kernels: apply_planes3 (one wave per SIMD, the default) and apply_planes4 (same records, half the
# LDS and registers per wave; mfgpu_desc.kernel = PLANES_2W)
PLANE_KERNELS = [(mf.KERNEL_PLANES, "apply_planes3"), (mf.KERNEL_PLANES_2W, "apply_planes4")]
PK = pytest.mark.parametrize("pk,pk_name", PLANE_KERNELS, ids=[k[1] for k in PLANE_KERNELS])

# (p, cells per direction): meshes with several batches of the default size (64 / n cells), ragged remainders included
SIZES = [(2, 5), (2, 9), (3, 4), (3, 7), (4, 3), (4, 5), (4, 7)]


@pytest.mark.parametrize("p,n", SIZES)
@pytest.mark.parametrize("nt", [mf.F64, mf.F32])
@PK
def test_planes_vmult_and_add(p, n, nt, pk, pk_name):
    mesh = mf.Mesh.uniform(3, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    mesh.desc.kernel = pk
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == pk_name
    st = op.plan_stats()
    assert st["max_batch_cells"] <= 64 // (p + 1) and st["n_batches"] >= 2
    rng = np.random.default_rng(1000 * p + n)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    xt = x.astype(mf.np_dtype(nt)).astype(np.float64)
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= TOL[nt]
    y0t = y0.astype(mf.np_dtype(nt)).astype(np.float64)
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= TOL[nt]


@pytest.mark.parametrize("p,n,cells", [(4, 4, 1), (4, 4, 2), (4, 4, 5), (4, 5, 7), (4, 6, 12), (3, 5, 3), (3, 6, 16),
                                       (2, 7, 4), (2, 8, 21)])
@PK
def test_planes_batch_sizes(p, n, cells, pk, pk_name):
    """batches of 1 .. 64/n cells (partly filled waves, idle lanes) give the same operator"""
    od = o.uniform_mesh_desc(3, p, n, coefficient=lambda x: 1.0 + x[..., 0] ** 2 + 0.5 * np.sin(3 * x[..., 1]) + x[..., 2])
    x = np.random.default_rng(p * 100 + n * 10 + cells).standard_normal(od.n_dofs)
    desc, keep = desc_from_oracle(od, max_cells_per_batch=cells, kernel=pk)
    op = mf.Operator(desc, keep)
    assert op.kernel_name() == pk_name
    assert op.plan_stats()["max_batch_cells"] <= cells
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12


def test_planes_agree_with_the_pencil_kernels():
    """same mesh, same vector through the three kernel families: each within 1e-12 of the oracle, and of each other"""
    mesh = mf.Mesh.uniform(3, 4, 6)
    od = oracle_desc_from_mesh(mesh)
    x = np.random.default_rng(5).standard_normal(mesh.n_dofs)
    ref = o.vmult(od, x)
    out = {}
    for kern, name in ((mf.KERNEL_PLANES, "apply_planes3"), (mf.KERNEL_PENCILS_X, "apply_batches_x"),
                       (mf.KERNEL_PENCILS, "apply_batches")):
        mesh.desc.kernel = kern
        op = mf.Operator(mesh.desc, mesh)
        assert op.kernel_name() == name
        out[name] = gpu_vmult(op, x)
        assert rel(out[name], ref) <= 1e-12
    assert rel(out["apply_planes3"], out["apply_batches_x"]) <= 1e-12


def test_planes_chained_applies_and_determinism():
    """bmop protocol (bmop.cu:134-146): dst = 0.1; repeat {swap; vmult}; bit-identical when repeated"""
    mesh = mf.Mesh.uniform(3, 4, 4)
    od = oracle_desc_from_mesh(mesh)
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == "apply_planes3"
    runs = []
    for _ in range(2):
        a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
        b.fill(0.1)
        for _ in range(3):
            a, b = b, a
            op.vmult(b, a)
        mf.synchronize()
        runs.append(b.to_host())
    np.testing.assert_array_equal(runs[0], runs[1])
    ref = o.bmop_protocol(od, 3)
    assert rel(runs[0], ref) <= 1e-12 * 100 ** 2


def test_planes_unsupported_requests_fail_loudly():
    mesh = mf.Mesh.uniform(2, 2, 4)
    mesh.desc.kernel = mf.KERNEL_PLANES
    with pytest.raises(mf.MfgpuError):
        mf.Operator(mesh.desc, mesh)


@pytest.mark.parametrize("p,n,wgs,kern,nt", [(4, 6, 1, mf.KERNEL_PLANES, mf.F64), (4, 7, 3, mf.KERNEL_PLANES, mf.F64),
                                             (4, 9, 5, mf.KERNEL_PLANES, mf.F32), (4, 12, 9, mf.KERNEL_PLANES, mf.F64),
                                             (3, 8, 3, mf.KERNEL_PLANES, mf.F64), (2, 9, 2, mf.KERNEL_PLANES, mf.F64),
                                             (4, 6, 1, mf.KERNEL_PLANES_2W, mf.F64), (4, 7, 3, mf.KERNEL_PLANES_2W, mf.F64),
                                             (4, 9, 5, mf.KERNEL_PLANES_2W, mf.F32), (3, 8, 3, mf.KERNEL_PLANES_2W, mf.F64),
                                             (2, 9, 2, mf.KERNEL_PLANES_2W, mf.F64),
                                             (4, 6, 2, mf.KERNEL_PENCILS_X, mf.F64), (2, 9, 3, mf.KERNEL_PENCILS_X, mf.F64),
                                             (6, 3, 1, mf.KERNEL_AUTO, mf.F64), (4, 5, 3, mf.KERNEL_PENCILS, mf.F64)])
def test_few_workgroups_walk_many_batches(p, n, wgs, kern, nt):
    """mfgpu_desc.max_workgroups: with a handful of workgroups on a small mesh every workgroup walks several batches,
    i.e. the persistent loop with its loads one and two batches ahead and its deferred scatter, which small meshes
    otherwise leave to the full-size property tests.  vmult, vmult_add and chained applies against the oracle."""
    mesh = mf.Mesh.uniform(3, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    mesh.desc.kernel = kern
    mesh.desc.max_workgroups = wgs
    mesh.desc.max_cells_per_batch = 0 if p == 4 else 8
    op = mf.Operator(mesh.desc, mesh)
    assert op.plan_stats()["n_batches"] >= 3 * wgs, op.plan_stats()
    rng = np.random.default_rng(17 * n + wgs)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    xt, y0t = (v.astype(mf.np_dtype(nt)).astype(np.float64) for v in (x, y0))
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= TOL[nt]
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= TOL[nt]
    if nt == mf.F64:
        a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
        b.fill(0.1)
        for _ in range(3):
            a, b = b, a
            op.vmult(b, a)
        mf.synchronize()
        assert rel(b.to_host(), o.bmop_protocol(od, 3)) <= 1e-12 * 100 ** 2


@pytest.mark.parametrize("make,kern", [(lambda: mf.Mesh.uniform(3, 4, 7), 0), (lambda: mf.Mesh.uniform(3, 2, 9), 0),
                                       (lambda: mf.Mesh.uniform(2, 4, 9), 0), (lambda: mf.Mesh.adaptive(3, 4, 4), 0),
                                       (lambda: mf.Mesh.ball(3, 2, 2), 0)])
def test_renumbered_mesh_gives_the_same_operator(make, kern):
    """optional batch-major dof numbering (mfgpu_suggest_renumbering + mfgpu_mesh_renumber): y_new[new(i)] = y_old[i]"""
    mesh = make()
    od0 = oracle_desc_from_mesh(mesh)
    x0 = np.random.default_rng(3).standard_normal(mesh.n_dofs)
    y0 = gpu_vmult(mf.Operator(mesh.desc, mesh), x0)
    assert rel(y0, o.vmult(od0, x0)) <= 1e-12
    ni = mesh.suggest_renumbering()
    mesh.renumber(ni)
    od1 = oracle_desc_from_mesh(mesh)
    x1 = np.empty_like(x0)
    x1[ni] = x0
    op = mf.Operator(mesh.desc, mesh)
    y1 = gpu_vmult(op, x1)
    assert rel(y1, o.vmult(od1, x1)) <= 1e-12
    assert rel(y1[ni], y0) <= 1e-12
    z = np.random.default_rng(4).standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x1, y0=z), o.vmult_add(od1, z, x1)) <= 1e-12


def _all_masks3():
    out = [t | (xyz << 3) for xyz in range(1, 8) for t in range(8)]
    for e in (1 << 6, 1 << 7, 1 << 8, (1 << 6) | (1 << 5), (1 << 7) | (1 << 3), (1 << 8) | (1 << 4)):
        out += [e | t for t in range(8)]
    return out


@pytest.mark.parametrize("p,n,wgs,nt", [(4, 5, 0, mf.F64), (4, 6, 2, mf.F64), (4, 5, 3, mf.F32), (3, 5, 0, mf.F64), (3, 6, 2, mf.F64),
                                        (2, 6, 0, mf.F64), (2, 7, 3, mf.F64)])
@PK
def test_planes_hanging_node_batches_synthetic_masks(p, n, wgs, nt, pk, pk_name):
    """apply_planes3<HN>: every mask of the reference's known-answer test (type x face bits, edge masks) on cells of a
    conforming mesh -- private entries, interpolation passes x, y, z before the cell stages and the transposed passes
    after them -- for every degree the plane kernel serves, few workgroups (several batches per workgroup) included;
    vmult and vmult_add against the oracle's emulation of resolve_hanging_nodes (hanging_nodes.cuh:617-778)"""
    od = o.uniform_mesh_desc(3, p, n)
    masks = _all_masks3()
    rng = np.random.default_rng(7 * p + n)
    cm = np.zeros(od.n_cells, dtype=np.uint32)
    pick = rng.permutation(od.n_cells)[:od.n_cells * 3 // 4]
    cm[pick] = np.array(masks, dtype=np.uint32)[np.arange(len(pick)) % len(masks)]
    od.constraint_mask = cm
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    xt, y0t = (v.astype(mf.np_dtype(nt)).astype(np.float64) for v in (x, y0))
    desc, keep = desc_from_oracle(od, number_type=nt, kernel=pk, max_workgroups=wgs)
    op = mf.Operator(desc, keep)
    assert op.kernel_name() == pk_name
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= TOL[nt]
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= TOL[nt]


@pytest.mark.parametrize("p,nref", [(4, 4), (4, 5), (3, 4), (2, 5)])
@PK
def test_planes_adaptive_mesh_and_inverse_diagonal(p, nref, pk, pk_name):
    """bmop -DADAPTIVE_GRID mesh entirely in the plane kernel (cells with a mask in batches of their own); the inverse
    diagonal runs on the same plan"""
    mesh = mf.Mesh.adaptive(3, p, nref)
    od = oracle_desc_from_mesh(mesh)
    mesh.desc.kernel = pk
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == pk_name
    rng = np.random.default_rng(p + nref)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12
    d = mf.DeviceVector(mesh.n_dofs)
    op.compute_inverse_diagonal(d)
    mf.synchronize()
    assert rel(d.to_host(), o.compute_inverse_diagonal(od)) <= 1e-12
