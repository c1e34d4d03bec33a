"""GPU parity tests: the HIP path, called through the C-ABI, against the oracle on the same inputs.

Tolerance (north_star / BASELINE.md section 2): relative l2 <= 1e-12 in double, <= 1e-5 in float,
after 1-3 applies."""
import glob
import os

import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from util import desc_from_oracle, oracle_desc_from_mesh

pytestmark = pytest.mark.gpu

TOL = {mf.F64: 1e-12, mf.F32: 1e-5}


def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / np.linalg.norm(b.astype(np.float64))


def gpu_vmult(op, x, number_type=mf.F64, y0=None):
    n = len(x)
    src, dst = mf.DeviceVector(n, number_type), mf.DeviceVector(n, number_type)
    src.from_host(x)
    if y0 is None:
        dst.fill(7.0)  # vmult must overwrite whatever is there (no separate dst = 0 pass)
        op.vmult(dst, src)
    else:
        dst.from_host(y0)
        op.vmult_add(dst, src)
    mf.synchronize()
    out = dst.to_host()
    np.testing.assert_array_equal(src.to_host(), x.astype(mf.np_dtype(number_type)))  # src untouched
    return out


CASES = [(2, 1, 7), (2, 2, 32), (2, 3, 5), (2, 4, 4), (2, 5, 3), (2, 6, 3),
         (3, 1, 5), (3, 2, 4), (3, 3, 3), (3, 4, 2), (3, 4, 5), (3, 5, 2), (3, 6, 2), (3, 6, 3)]


@pytest.mark.parametrize("dim,p,n", CASES)
@pytest.mark.parametrize("nt", [mf.F64, mf.F32])
@pytest.mark.parametrize("colored", [False, True])
def test_vmult_matches_oracle(dim, p, n, nt, colored):
    mesh = mf.Mesh.uniform(dim, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    if colored:
        mesh.desc.flags |= mf.COLORED_SCATTER
    op = mf.Operator(mesh.desc, mesh)
    assert op.n() == mesh.n_dofs
    x = np.random.default_rng(dim * 100 + p * 10 + n).standard_normal(mesh.n_dofs)
    y = gpu_vmult(op, x, nt)
    assert rel(y, o.vmult(od, x.astype(mf.np_dtype(nt)).astype(np.float64))) <= TOL[nt]


KERNELS = [(mf.KERNEL_PENCILS, "apply_batches"), (mf.KERNEL_PENCILS_X, "apply_batches_x"),
           (mf.KERNEL_PLANES, "apply_planes3"), (mf.KERNEL_PLANES_2W, "apply_planes4")]


@pytest.mark.parametrize("kern,name", KERNELS, ids=[k[1] for k in KERNELS])
@pytest.mark.parametrize("p,n", [(4, 7), (2, 9), (3, 6)])
@pytest.mark.parametrize("nt", [mf.F64, mf.F32])
def test_kernel_families_match_oracle(kern, name, p, n, nt):
    """every cell-loop kernel family that covers a 3D conforming mesh (mfgpu_desc.kernel) computes the same
    operator: vmult and vmult_add against the oracle"""
    mesh = mf.Mesh.uniform(3, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    mesh.desc.kernel = kern
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == name
    rng = np.random.default_rng(17)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    xt, y0t = (v.astype(mf.np_dtype(nt)).astype(np.float64) for v in (x, y0))
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= TOL[nt]
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= TOL[nt]


# batches of several chunks of cells (the production default on large meshes; small meshes default to one cell per
# batch): degree-dependent lane masks, idle lanes and chunk-boundary clamps of the pencil kernels
@pytest.mark.parametrize("p,n,cells", [(1, 12, 256), (2, 10, 256), (3, 8, 256), (4, 6, 27), (5, 4, 256), (6, 3, 256)])
@pytest.mark.parametrize("kern", [mf.KERNEL_PENCILS, mf.KERNEL_PENCILS_X], ids=["apply_batches", "apply_batches_x"])
@pytest.mark.parametrize("nt", [mf.F64, mf.F32])
def test_pencil_kernels_multi_chunk_batches(p, n, cells, kern, nt):
    mesh = mf.Mesh.uniform(3, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    mesh.desc.kernel = kern
    mesh.desc.max_cells_per_batch = cells
    op = mf.Operator(mesh.desc, mesh)
    st = op.plan_stats()
    assert st["n_batches"] >= 2 and st["max_batch_cells"] > 256 // (p + 1) ** 2, st  # more than one chunk
    rng = np.random.default_rng(p * 31 + n)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    xt, y0t = (v.astype(mf.np_dtype(nt)).astype(np.float64) for v in (x, y0))
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= TOL[nt]
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= TOL[nt]


@pytest.mark.parametrize("colored", [False, True])
@pytest.mark.parametrize("dim,p,n", [(2, 2, 9), (3, 4, 3), (3, 2, 5)])
def test_vmult_add_matches_oracle(dim, p, n, colored):
    mesh = mf.Mesh.uniform(dim, p, n)
    od = oracle_desc_from_mesh(mesh)
    if colored:
        mesh.desc.flags |= mf.COLORED_SCATTER
    op = mf.Operator(mesh.desc, mesh)
    rng = np.random.default_rng(11)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12


def test_three_way_differential_reference_config():
    """test_laplace_op.cu on the GPU path: [0,1]^2, p=4, 4x4 cells: GPU-MF vs CPU-MF vs assembled."""
    mesh = mf.Mesh.uniform(2, 4, 4, lo=0.0, hi=1.0)
    od = oracle_desc_from_mesh(mesh)
    op = mf.Operator(mesh.desc, mesh)
    x = np.random.default_rng(1).random(289)
    y_gpu = gpu_vmult(op, x)
    y_cpu = o.vmult(od, x)
    y_sp = o.assemble(od) @ x
    assert rel(y_gpu, y_cpu) <= 1e-12 and rel(y_gpu, y_sp) <= 1e-12 and rel(y_cpu, y_sp) <= 1e-12


def test_independent_oracle_mesh_and_given_coefficient():
    """description built by the ORACLE's own mesh code, explicit coefficient array (not evaluated on
    the device), non-default batch limits"""
    od = o.uniform_mesh_desc(3, 4, 4, coefficient=lambda x: 1.0 + x[..., 0] ** 2 + 0.5 * np.sin(x[..., 1]))
    x = np.random.default_rng(3).standard_normal(od.n_dofs)
    ref = o.vmult(od, x)
    for kw in ({}, dict(max_cells_per_batch=1), dict(max_cells_per_batch=8), dict(max_cells_per_batch=64, max_dofs_per_batch=4000),
               dict(colored=True), dict(colored=True, max_cells_per_batch=8)):
        desc, keep = desc_from_oracle(od, **kw)
        op = mf.Operator(desc, keep)
        assert rel(gpu_vmult(op, x), ref) <= 1e-12, kw


GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_golden_fixtures_and_bmop_protocol(path):
    """committed fixtures: one apply on a fixed random vector + bmop's protocol (bmop.cu:137-146):
    dst = 0.1; k x { swap; vmult }"""
    g = np.load(path)
    mesh = mf.Mesh.uniform(int(g["dim"]), int(g["degree"]), int(g["n"]))
    op = mf.Operator(mesh.desc, mesh)
    assert rel(gpu_vmult(op, g["x"]), g["y"]) <= 1e-12
    n = mesh.n_dofs
    dst, src = mf.DeviceVector(n), mf.DeviceVector(n)
    dst.fill(0.1)
    prev = np.full(n, 0.1)
    for k in (1, 2, 3):
        dst.swap(src)
        op.vmult(dst, src)
        mf.synchronize()
        # chained applies of the un-normalised operator amplify rounding-level differences of the
        # previous iterate by ||A|| per apply ("parity under reordering"): the
        # chained iterate is held to 1e-12 * 100^(k-1); every SINGLE apply is held to 1e-12 below.
        assert rel(dst.to_host(), g[f"prot{k}"]) <= 1e-12 * 100 ** (k - 1), k
        assert rel(gpu_vmult(op, prev), g[f"prot{k}"]) <= 1e-12, k  # one apply on the oracle's iterate
        prev = g[f"prot{k}"]


def test_ragged_mesh_with_orphans():
    od = o.uniform_mesh_desc(2, 2, 4)
    keep_cells = np.array([0, 1, 2, 5, 10, 15])
    od2 = o.Desc(2, 2, od.n_dofs, od.loc2glob[keep_cells], od.JxW[keep_cells], od.inv_jac[keep_cells],
                 od.coefficient[keep_cells], od.constrained)
    for colored in (False, True):
        desc, keep = desc_from_oracle(od2, max_cells_per_batch=3, colored=colored)
        op = mf.Operator(desc, keep)
        assert op.plan_stats()["n_orphans"] > 0
        x = np.random.default_rng(1).standard_normal(od.n_dofs)
        assert rel(gpu_vmult(op, x), o.vmult(od2, x)) <= 1e-12
        y0 = np.random.default_rng(2).standard_normal(od.n_dofs)
        assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od2, y0, x)) <= 1e-12


def test_errors_are_loud():
    mesh = mf.Mesh.uniform(2, 2, 3)
    op = mf.Operator(mesh.desc, mesh)
    v = mf.DeviceVector(mesh.n_dofs)
    with pytest.raises(mf.MfgpuError, match="alias"):
        op.vmult(v, v)
    d2 = mf.Desc.from_buffer_copy(mesh.desc)
    d2.flags = mf.COLORED_SCATTER  # general-Jacobian path (no UNIFORM_J0): two-pass scatter mode only, no fallback
    with pytest.raises(mf.MfgpuError, match="UNIFORM_J0"):
        mf.Operator(d2, mesh)


@pytest.mark.parametrize("p,n,nt", [(4, 54, mf.F64), (6, 36, mf.F64), (4, 54, mf.F32)], ids=["C2_p4_n54", "C5_p6_n36", "C2_float"])
def test_full_size_matches_cpu_twin(p, n, nt):
    """BASELINE configs C2 (p = 4, n = 54: 10 218 313 dofs) and C5 (p = 6, n = 36) at FULL size against the CPU twin
    oracle/cpu_ref.c (pinned to the numpy oracle by tests/test_cpu_ref.py; the restatement of
    laplace_operator_cpu.cc:122-143, 178-211) on the same mesh and the same seeded random vector -- every cell
    contributes, unlike the protocol's vector of 0.1s.  Tolerance: relative l2 <= 1e-12 (double), 1e-5 (float)."""
    from oracle import cpu_ref
    mesh = mf.Mesh.uniform(3, p, n, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    ref = cpu_ref.CpuRef(od, cpu_ref.structured_cell_colors([n] * 3))
    cpu_ref.set_threads(cpu_ref.cpu_share())
    op = mf.Operator(mesh.desc, mesh)
    rng = np.random.default_rng(p * 1000 + n)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    xt, y0t = (v.astype(mf.np_dtype(nt)).astype(np.float64) for v in (x, y0))
    want = ref.vmult(xt)
    assert rel(gpu_vmult(op, x, nt), want) <= TOL[nt]
    assert rel(gpu_vmult(op, x, nt, y0=y0), want + y0t) <= TOL[nt]
    if p == 4 and nt == mf.F64:  # the other plane kernel on the same inputs
        mesh.desc.kernel = mf.KERNEL_PLANES_2W
        assert rel(gpu_vmult(mf.Operator(mesh.desc, mesh), x, nt), want) <= TOL[nt]


@pytest.mark.parametrize("n", [54])
def test_full_size_properties(n):
    """BASELINE config C2 (p=4, 3D, n=54: 157 464 cells, 10 218 313 dofs): size-independent
    properties instead of the oracle: linearity, symmetry, constants in the kernel, identity rows."""
    mesh = mf.Mesh.uniform(3, 4, n)
    N = mesh.n_dofs
    assert N == 217 ** 3
    op = mf.Operator(mesh.desc, mesh)
    st = op.plan_stats()
    assert st["n_launches"] == 1
    con = mesh.arrays()["constrained_dofs"]
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(N), rng.standard_normal(N)
    Au, Av = gpu_vmult(op, u), gpu_vmult(op, v)
    # identity rows
    np.testing.assert_array_equal(Au[con], u[con])
    # symmetry (on the free block; constrained rows are identity)
    uf, vf = u.copy(), v.copy()
    uf[con] = 0
    vf[con] = 0
    Auf, Avf = gpu_vmult(op, uf), gpu_vmult(op, vf)
    assert abs(vf @ Auf - uf @ Avf) <= 1e-11 * abs(vf @ Auf)
    assert uf @ Auf > 0
    # linearity
    w = 0.3 * u - 1.7 * v
    assert rel(gpu_vmult(op, w), 0.3 * Au - 1.7 * Av) <= 1e-12
    # constants: rows away from the boundary vanish
    y = gpu_vmult(op, np.ones(N))
    g = 217
    idx = np.arange(N)
    ix, iy, iz = idx % g, (idx // g) % g, idx // (g * g)
    deep = (ix >= 5) & (ix <= g - 6) & (iy >= 5) & (iy <= g - 6) & (iz >= 5) & (iz <= g - 6)
    assert np.abs(y[deep]).max() <= 1e-10 * np.abs(y).max()
    # a z-slab of the same mesh reproduces the interior rows of the full operator (multi-GPU shards)
    slab = mf.Mesh.uniform(3, 4, n, slab=(10, 14))
    ops = mf.Operator(slab.desc, slab)
    gz0 = 10 * 4 * g * g
    us = u[gz0:gz0 + slab.n_dofs]
    ys = gpu_vmult(ops, us)
    lo, hi = g * g, slab.n_dofs - g * g  # rows not on the two slab interface planes
    free = np.ones(slab.n_dofs, bool)
    free[slab.arrays()["constrained_dofs"]] = False
    sel = np.zeros(slab.n_dofs, bool)
    sel[lo:hi] = True
    sel &= free
    # full-mesh rows strictly inside the slab only see slab cells, but src entries at constrained dofs
    # differ (zeroed) identically in both; compare directly
    assert rel(ys[sel], Au[gz0:gz0 + slab.n_dofs][sel]) <= 1e-12


def _all_masks(dim):
    """every mask of the reference's known-answer test (type bits x face bits) + 3D edge masks"""
    out = []
    for xyz in range(1, (1 << dim) - 1 + 1):
        for t in range(1 << dim):
            out.append(t | (xyz << 3))
    if dim == 3:
        for e in (1 << 6, 1 << 7, 1 << 8, (1 << 6) | (1 << 5), (1 << 7) | (1 << 3), (1 << 8) | (1 << 4)):
            for t in range(8):
                out.append(e | t)
    return out


@pytest.mark.parametrize("dim,p,n", [(2, 1, 8), (2, 2, 8), (2, 4, 8), (3, 1, 5), (3, 2, 4), (3, 3, 4), (3, 4, 4), (3, 5, 3)])
@pytest.mark.parametrize("colored", [False, True, "x"])
def test_hanging_node_stages_synthetic_masks(dim, p, n, colored):
    """in-kernel resolve_hanging_nodes (NOTRANSPOSE before evaluate, TRANSPOSE after integrate,
    fee_gpu.cuh:333-335,349-351) against the oracle's emulation for every mask type.  The masks are
    assigned to cells of a conforming mesh: algebraically A = sum_cells P^T C^T K C P either way."""
    if colored == "x":  # two-pass mode with apply_batches instead of the 3D default apply_batches_x
        if dim != 3:
            pytest.skip("apply_batches_x is a 3D kernel: 2D always runs apply_batches")
        colored = False
        kernel = mf.KERNEL_PENCILS
    else:
        kernel = mf.KERNEL_AUTO
    od = o.uniform_mesh_desc(dim, p, n)
    masks = _all_masks(dim)
    rng = np.random.default_rng(5)
    cm = np.zeros(od.n_cells, dtype=np.uint32)
    pick = rng.permutation(od.n_cells)[:min(od.n_cells * 2 // 3, 4 * len(masks))]
    cm[pick] = np.array(masks, dtype=np.uint32)[np.arange(len(pick)) % len(masks)]
    od.constraint_mask = cm
    x = rng.standard_normal(od.n_dofs)
    ref = o.vmult(od, x)
    desc, keep = desc_from_oracle(od, colored=colored, kernel=kernel)
    op = mf.Operator(desc, keep)
    assert rel(gpu_vmult(op, x), ref) <= 1e-12
    # and the assembled C^T K C form
    assert rel(ref, o.assemble(od) @ x) <= 1e-12


@pytest.mark.parametrize("dim,p,nref", [(2, 2, 4), (2, 4, 5), (3, 1, 4), (3, 2, 4), (3, 4, 4), (3, 4, 5), (3, 3, 5)])
@pytest.mark.parametrize("colored", [False, True])
def test_adaptive_mesh_with_hanging_nodes(dim, p, nref, colored):
    """BASELINE configs[2]: bmop -DADAPTIVE_GRID recipe (bmop_common.h:49-105) with hanging-node
    constraints handled in the kernel; GPU vs the oracle's emulation of the reference GPU path."""
    mesh = mf.Mesh.adaptive(dim, p, nref)
    assert (mesh.arrays()["constraint_mask"] != 0).any()
    od = oracle_desc_from_mesh(mesh)
    if colored:
        mesh.desc.flags |= mf.COLORED_SCATTER
    op = mf.Operator(mesh.desc, mesh)
    assert op.plan_stats()["n_orphans"] > 0  # the hanging dofs
    rng = np.random.default_rng(dim + p + nref)
    x = rng.standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    y0 = rng.standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12


@pytest.mark.parametrize("p,nref", [(4, 4), (2, 4), (3, 5)])
@pytest.mark.parametrize("xk", [True, False])
def test_adaptive_mesh_both_cell_loop_kernels(p, nref, xk):
    """3D two-pass default on meshes with hanging nodes: apply_planes3 at p = 4, apply_planes4 at p = 3 (cells with a mask
    in batches of their own, <HN>), apply_batches_x otherwise; mfgpu_desc.kernel = PENCILS selects apply_batches (which also
    serves 2D and the coloured mode), PENCILS_X the pencil kernel at p = 4 too: same operator on an adaptive mesh."""
    mesh = mf.Mesh.adaptive(3, p, nref)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    mesh.desc.kernel = mf.KERNEL_AUTO if xk else mf.KERNEL_PENCILS
    op = mf.Operator(mesh.desc, mesh)
    auto = {4: "apply_planes3", 3: "apply_planes4", 2: "apply_batches_x"}[p]  # (p = 3 in double: apply_planes4)
    assert op.kernel_name() == (auto if xk else "apply_batches")
    # batches of masked and unmasked cells interleaved, ONE instantiation: one launch (p = 4: two halves, pass 2 of the
    # first beside the second); p = 2: plane-less, the pencil kernel in one launch
    assert op.plan_stats()["n_launches"] == (2 if (p == 4 and xk) else 1)
    rng = np.random.default_rng(5)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12


@pytest.mark.parametrize("p,nref,nt", [(5, 4, mf.F64), (6, 4, mf.F64), (5, 4, mf.F32)])
def test_adaptive_mesh_high_degree_two_families(p, nref, nt):
    """p = 5, 6 on a mesh with hanging nodes: the cells without a constraint mask run in the plane kernel
    (apply_planes4w, one wave per SIMD), the masked ones in the pencil kernel's hanging-node variant; vmult and
    vmult_add against the oracle"""
    mesh = mf.Mesh.adaptive(3, p, nref, number_type=nt)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == "apply_planes4+apply_batches_x"
    assert op.plan_stats()["n_launches"] == 2  # the two families, pass 2 of the first's dofs beside the second
    rng = np.random.default_rng(p)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    xt, y0t = (v.astype(mf.np_dtype(nt)).astype(np.float64) for v in (x, y0))
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= TOL[nt]
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= TOL[nt]


@pytest.mark.parametrize("p,nref", [(4, 4), (4, 5), (2, 5), (3, 5)])
@pytest.mark.parametrize("kern", [mf.KERNEL_AUTO, mf.KERNEL_PENCILS_X, mf.KERNEL_PENCILS],
                         ids=["auto", "apply_batches_x", "apply_batches"])
def test_adaptive_mesh_multi_chunk_batches(p, nref, kern):
    """hanging-node batches of several chunks of cells (max_cells_per_batch = 64: the production size on large
    meshes) in every kernel family that handles them, against the oracle"""
    mesh = mf.Mesh.adaptive(3, p, nref)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    mesh.desc.kernel = kern
    mesh.desc.max_cells_per_batch = 64
    op = mf.Operator(mesh.desc, mesh)
    if not (kern == mf.KERNEL_AUTO and p >= 3):  # (the plane kernel's batches are one wave: 64 / n cells)
        assert op.plan_stats()["max_batch_cells"] > 256 // (p + 1) ** 2
    rng = np.random.default_rng(p + nref)
    x, y0 = rng.standard_normal(mesh.n_dofs), rng.standard_normal(mesh.n_dofs)
    assert rel(gpu_vmult(op, x), o.vmult(od, x)) <= 1e-12
    assert rel(gpu_vmult(op, x, y0=y0), o.vmult_add(od, y0, x)) <= 1e-12


def test_c5_full_size_properties():
    """BASELINE configs[4] at full size (p = 6, 3D, 36^3 cells, 217^3 dofs): linearity, symmetry, identity rows,
    constants in the kernel -- the size-independent properties of test_full_size_properties"""
    mesh = mf.Mesh.uniform(3, 6, 36)
    N = mesh.n_dofs
    assert N == 217 ** 3
    op = mf.Operator(mesh.desc, mesh)
    con = mesh.arrays()["constrained_dofs"]
    rng = np.random.default_rng(6)
    u, v = rng.standard_normal(N), rng.standard_normal(N)
    u[con] = 0
    v[con] = 0
    Au, Av = gpu_vmult(op, u), gpu_vmult(op, v)
    assert np.abs(Au[con]).max() == 0.0                           # identity rows
    assert abs(v @ Au - u @ Av) <= 1e-11 * abs(v @ Au) and u @ Au > 0
    assert rel(gpu_vmult(op, 0.3 * u - 1.7 * v), 0.3 * Au - 1.7 * Av) <= 1e-12
    y = gpu_vmult(op, np.ones(N))
    g = 217
    idx = np.arange(N)
    ix, iy, iz = idx % g, (idx // g) % g, idx // (g * g)
    deep = (ix >= 7) & (ix <= g - 8) & (iy >= 7) & (iy <= g - 8) & (iz >= 7) & (iz <= g - 8)
    assert np.abs(y[deep]).max() <= 1e-10 * np.abs(y).max()


def test_adaptive_float():
    mesh = mf.Mesh.adaptive(3, 4, 4, number_type=mf.F32)
    od = oracle_desc_from_mesh(mesh, dtype=np.float64)
    op = mf.Operator(mesh.desc, mesh)
    x = np.random.default_rng(9).standard_normal(mesh.n_dofs).astype(np.float32)
    assert rel(gpu_vmult(op, x, mf.F32), o.vmult(od, x.astype(np.float64))) <= 1e-5


def test_adaptive_full_size_properties():
    """configs[2] at the reference's scale (n_ref = 6): linearity, symmetry, identity rows"""
    mesh = mf.Mesh.adaptive(3, 4, 6)
    N = mesh.n_dofs
    op = mf.Operator(mesh.desc, mesh)
    con = mesh.arrays()["constrained_dofs"]
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(N), rng.standard_normal(N)
    Au, Av = gpu_vmult(op, u), gpu_vmult(op, v)
    np.testing.assert_array_equal(Au[con], u[con])
    uf, vf = u.copy(), v.copy()
    uf[con] = 0
    vf[con] = 0
    Auf, Avf = gpu_vmult(op, uf), gpu_vmult(op, vf)
    assert abs(vf @ Auf - uf @ Avf) <= 1e-11 * abs(vf @ Auf)
    assert rel(gpu_vmult(op, 0.5 * u + 2.0 * v), 0.5 * Au + 2.0 * Av) <= 1e-12


def test_bmop_driver_binaries():
    """C++ shim + bmop driver (dealii-cuda_amd/host): same CLI and TSV line as reference bmop.cu:152,186-192.
    C1 plumbing config: DEGREE_FE=2, DIMENSION=2, 5 global refinements -> 4225 dofs."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = os.path.join(root, "dealii-cuda_amd", "host", "bin")
    out = subprocess.run([os.path.join(b, "bmop-2d-p2"), "5", "4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [ln.split("\t") for ln in out.stdout.strip().splitlines()]
    assert [ln[:3] for ln in lines] == [["2", "2", "1089"], ["2", "2", "4225"]]
    assert all(float(ln[3]) > 0 for ln in lines)
    out = subprocess.run([os.path.join(b, "bmop-3d-p4"), "4", "4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split("\t")[:3] == ["3", "4", str(65 ** 3)]
    out = subprocess.run([os.path.join(b, "bmop-3d-p4-adaptive"), "4", "4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split("\t")[:2] == ["3", "4"]
    # -DBALL_GRID: hyper_ball, 3 global refinements = 7 * 8^3 cells, 232609 dofs at p = 4
    out = subprocess.run([os.path.join(b, "bmop-3d-p4-ball"), "3", "3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split("\t")[:3] == ["3", "4", "232609"]
    # 2D ball: 5 * 4^5 cells; p = 2: 20609 dofs (vertices + lines + cells)
    out = subprocess.run([os.path.join(b, "bmop-2d-p2-ball"), "5", "5"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split("\t")[:3] == ["2", "2", "20609"]


@pytest.mark.parametrize("p,n,world", [(4, 7, 2), (2, 9, 3), (4, 12, 4)])
def test_slab_decomposition_with_the_gpu_operator(p, n, world):
    """Row (e) on the device, without a second GPU: the HIP operator applied slab by slab (z-slab meshes of
    the multi-GPU mode, interface planes that are NOT boundary), the interface planes summed as
    SlabExchange does, against the single-domain oracle.  Two chained applies: the summed planes must be
    consistent ghost values for the next apply."""
    from pymfgpu.parallel import slab_ranges

    full = mf.Mesh.uniform(3, p, n)
    od = oracle_desc_from_mesh(full, dtype=np.float64)
    fxyz = full.dof_coords()
    key = {tuple(np.round(c, 9)): i for i, c in enumerate(fxyz)}
    src_g = np.sin(3 * fxyz[:, 0]) + fxyz[:, 1] ** 2 - np.cos(2 * fxyz[:, 2]) * fxyz[:, 0]
    slabs = []
    for zb, ze in slab_ranges(n, world):
        mesh = mf.Mesh.uniform(3, p, n, slab=(zb, ze))
        gi = np.array([key[tuple(np.round(c, 9))] for c in mesh.dof_coords()])
        slabs.append((mesh, mf.Operator(mesh.desc, mesh), gi))
    con = np.zeros(full.n_dofs, bool)
    con[od.constrained] = True
    x = src_g
    for _ in range(2):
        acc = np.zeros(full.n_dofs)
        cnt = np.zeros(full.n_dofs, int)
        for mesh, op, gi in slabs:
            y = gpu_vmult(op, x[gi])
            np.add.at(acc, gi, y)
            np.add.at(cnt, gi, 1)
        assert cnt.max() == 2 and cnt.min() == 1           # only interface planes are shared, by two slabs
        acc[con & (cnt == 2)] *= 0.5                        # identity rows on both sides: not summed
        ref = o.vmult(od, x)
        assert rel(acc, ref) <= 1e-12
        x = acc


@pytest.mark.parametrize("p,n,world,first", [(4, 6, 2, None), (4, 7, 3, None), (2, 8, 4, None), (3, 6, 2, None),
                                             (4, 12, 2, [True, True]), (4, 18, 3, [True, False, True]), (4, 24, 3, [True, True, True]),
                                             (2, 18, 3, [True, True, True]), (3, 24, 3, "hole"), (5, 24, 3, "hole"),
                                             (6, 24, 3, "hole")])
@pytest.mark.parametrize("nt", [mf.F64, mf.F32])
def test_cxx_slab_exchange_in_process(p, n, world, first, nt):
    """the C++ multi-GPU path behind the C-ABI (mfgpu_dist: priority pass 2 of the interface planes, pack, transfer,
    masked add, overlap with the rest of pass 2) on ONE GPU: every slab is an operator + mfgpu_dist of this process,
    connected by the in-process transport -- everything but the two RCCL calls of the real transport.  Three chained
    distributed applies against the single-domain oracle."""
    from pymfgpu.parallel import slab_ranges

    full = mf.Mesh.uniform(3, p, n)
    od = oracle_desc_from_mesh(full, dtype=np.float64)
    key = {tuple(np.round(c, 9)): i for i, c in enumerate(full.dof_coords())}
    slabs = []
    for r, (zb, ze) in enumerate(slab_ranges(n, world)):
        mesh = mf.Mesh.uniform(3, p, n, slab=(zb, ze), number_type=nt)
        gi = np.array([key[tuple(np.round(c, 9))] for c in mesh.dof_coords()])
        op = mf.Operator(mesh.desc, mesh)
        dist = mf.Dist(mesh, r, world)
        dist.attach(op)
        a, b = mf.DeviceVector(mesh.n_dofs, nt), mf.DeviceVector(mesh.n_dofs, nt)
        slabs.append(dict(mesh=mesh, op=op, dist=dist, gi=gi, a=a, b=b))
    for lo, up in zip(slabs, slabs[1:]):
        lo["dist"].connect_local(up["dist"])
    # on slabs thick enough the batches touching an interface plane run first and the exchange
    # runs beside the interior batches; the interface batches sit at the ends of the plan's batch order
    for r, s in enumerate(slabs):
        ifirst, r1, r2, nb = s["dist"].schedule()
        if first == "hole":  # the middle slab's two interface ranges run as ONE launch with a hole (both plane kernels)
            assert ifirst and (r != 1 or 0 < r1 < r2 < nb), (r, ifirst, r1, r2, nb)
        elif first is not None:
            assert ifirst == first[r], (r, ifirst, r1, r2, nb)
        if ifirst:
            assert 0 <= r1 < r2 <= nb and (r2 - r1) * 10 >= nb * 4
            assert (r1 == 0) == (r == 0) and (r2 == nb) == (r == world - 1)
        with pytest.raises(mf.MfgpuError):  # one-call form on the in-process transport: peers have not packed yet
            s["dist"].vmult(s["op"], s["b"], s["a"])
    fx = full.dof_coords()
    x = (np.sin(3 * fx[:, 0]) + fx[:, 1] ** 2 - np.cos(2 * fx[:, 2]) * fx[:, 0]).astype(mf.np_dtype(nt)).astype(np.float64)
    for s in slabs:
        s["a"].from_host(x[s["gi"]])
    ref = x
    for it in range(3):
        for s in slabs:
            s["dist"].vmult_begin(s["op"], s["b"], s["a"])
        for s in slabs:
            s["dist"].vmult_end(s["op"], s["b"])
        mf.synchronize()
        ref = o.vmult(od, ref)
        for s in slabs:
            assert rel(s["b"].to_host(), ref[s["gi"]]) <= TOL[nt] * 100 ** it
            s["a"], s["b"] = s["b"], s["a"]
    # one slab, no neighbours: the distributed entry point is the plain vmult
    mesh = mf.Mesh.uniform(3, p, n, number_type=nt)
    op, dist = mf.Operator(mesh.desc, mesh), mf.Dist(mesh, 0, 1)
    dist.attach(op)
    a, b = mf.DeviceVector(mesh.n_dofs, nt), mf.DeviceVector(mesh.n_dofs, nt)
    a.from_host(x)
    dist.vmult(op, b, a)
    mf.synchronize()
    assert rel(b.to_host(), o.vmult(od, x)) <= TOL[nt]


def test_create_destroy_returns_all_device_memory():
    """mfgpu_destroy / mfgpu_vec_free give back everything mfgpu_create / mfgpu_vec_alloc / the first vmult,
    diagonal and reduction took: repeated set-up and tear-down of operators of every kernel family must not
    shrink the free device memory (hipMemGetInfo)."""
    from util import deformed_oracle_desc

    def cycle():
        for kind in ("uniform", "adaptive", "colored", "2d", "general"):
            if kind == "general":
                desc, keep = desc_from_oracle(deformed_oracle_desc(2, 4))
            else:
                keep = (mf.Mesh.adaptive(3, 2, 4) if kind == "adaptive" else
                        mf.Mesh.uniform(2, 3, 20) if kind == "2d" else mf.Mesh.uniform(3, 4, 8))
                desc = keep.desc
                if kind == "colored":
                    desc.flags |= mf.COLORED_SCATTER
            op = mf.Operator(desc, keep)
            n = op.n()
            a, b = mf.DeviceVector(n), mf.DeviceVector(n)
            a.fill(1.0)
            op.vmult(b, a)
            op.compute_inverse_diagonal(a)
            b.dot(a)
            mf.synchronize()
            op.clear()
            del a, b, op

    cycle()                       # first use allocates the per-device reduction scratch (kept by design)
    mf.synchronize()
    free0, total = mf.device_memory_info()
    for _ in range(5):
        cycle()
    mf.synchronize()
    free1, _ = mf.device_memory_info()
    assert free0 - free1 <= 8 << 20, (free0, free1)   # allocator granularity only, no growth per cycle
