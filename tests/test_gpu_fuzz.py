"""Seeded random configurations of the operator against the oracle: dimension, degree, mesh size, hanging-node masks on
random cells, number type, scatter mode, kernel family, batch limits, cell-loop segments, workgroup cap -- the
combinations the hand-written cases do not enumerate.  Unsupported combinations must fail loudly (MfgpuError), never
give a wrong result.  Tolerance: relative l2 <= 1e-12 (double), 1e-5 (float)."""
import numpy as np
import pytest

import pymfgpu as mf
from oracle import mf_oracle as o
from test_gpu import TOL, gpu_vmult, rel
from util import desc_from_oracle

pytestmark = pytest.mark.gpu
_ran, _refused = [], []


def _masks(dim):
    if dim == 2:
        return [t | (xy << 3) for xy in (1, 2) for t in range(4)]
    out = [t | (xyz << 3) for xyz in range(1, 8) for t in range(8)]
    for e in (1 << 6, 1 << 7, 1 << 8, (1 << 6) | (1 << 5), (1 << 7) | (1 << 3), (1 << 8) | (1 << 4)):
        out += [e | t for t in range(8)]
    return out


@pytest.mark.parametrize("seed", range(400))
def test_random_configuration(seed):
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([2, 3, 3, 3]))
    p = int(rng.integers(1, 7)) if dim == 2 else int(rng.choice([1, 2, 2, 3, 3, 4, 4, 4, 5, 6]))
    n = int(rng.integers(2, {1: 9, 2: 8, 3: 6, 4: 6, 5: 4, 6: 3}[p] + 1)) if dim == 3 else int(rng.integers(2, 12))
    nt = mf.F32 if rng.random() < 0.25 else mf.F64
    od = o.uniform_mesh_desc(dim, p, n)
    if rng.random() < 0.5:
        masks = _masks(dim)
        cm = np.zeros(od.n_cells, dtype=np.uint32)
        pick = rng.permutation(od.n_cells)[:int(od.n_cells * rng.random())]
        cm[pick] = rng.choice(np.array(masks, dtype=np.uint32), size=len(pick))
        od.constraint_mask = cm
    kw = dict(
        colored=bool(rng.random() < 0.2),
        kernel=int(rng.choice([mf.KERNEL_AUTO, mf.KERNEL_AUTO, mf.KERNEL_PENCILS, mf.KERNEL_PENCILS_X, mf.KERNEL_PLANES])),
        max_cells_per_batch=int(rng.choice([0, 0, 1, 2, 5, 8, 12, 27, 64])),
        cell_loop_segments=int(rng.choice([0, 0, 1, 2, 3])),
        max_workgroups=int(rng.choice([0, 0, 1, 2, 7])),
    )
    x, y0 = rng.standard_normal(od.n_dofs), rng.standard_normal(od.n_dofs)
    xt, y0t = (v.astype(mf.np_dtype(nt)).astype(np.float64) for v in (x, y0))
    desc, keep = desc_from_oracle(od, number_type=nt, **kw)
    try:
        op = mf.Operator(desc, keep)
    except mf.MfgpuError as e:
        _refused.append((seed, str(e)))
        return  # a combination the library refuses (e.g. PLANES in 2D): loud, not wrong
    _ran.append(seed)
    info = (seed, dim, p, n, nt, kw, op.kernel_name())
    assert rel(gpu_vmult(op, x, nt), o.vmult(od, xt)) <= TOL[nt], info
    assert rel(gpu_vmult(op, x, nt, y0=y0), o.vmult_add(od, y0t, xt)) <= TOL[nt], info


def test_most_random_configurations_ran():
    """(runs after the cases above) the refusals are the documented ones, and most configurations computed"""
    assert len(_ran) >= 75, (_ran, _refused)
    for _, msg in _refused:
        assert "kernel" in msg or "UNIFORM_J0" in msg or "two-pass" in msg, msg
