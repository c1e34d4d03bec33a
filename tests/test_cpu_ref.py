"""oracle/cpu_ref.c (the timed CPU baseline) against the numpy oracle and the golden fixtures."""
import glob
import os

import numpy as np
import pytest

from oracle import cpu_ref
from oracle import mf_oracle as o


@pytest.mark.parametrize("dim,p,n", [(2, 2, 8), (2, 4, 4), (3, 1, 4), (3, 4, 3), (3, 6, 2)])
def test_cpu_ref_matches_numpy_oracle(dim, p, n):
    od = o.uniform_mesh_desc(dim, p, n)
    colors = cpu_ref.structured_cell_colors([n] * dim)
    np.testing.assert_array_equal(np.sort(np.unique(colors)), np.arange(2 ** dim))
    ref = cpu_ref.CpuRef(od, colors)
    x = np.random.default_rng(4).standard_normal(od.n_dofs)
    y = ref.vmult(x)
    yo = o.vmult(od, x)
    assert np.linalg.norm(y - yo) <= 1e-13 * np.linalg.norm(yo)
    # generic colouring gives the same result up to summation order
    ref2 = cpu_ref.CpuRef(od, cpu_ref.greedy_cell_colors(od.loc2glob, od.n_dofs))
    assert np.linalg.norm(ref2.vmult(x) - yo) <= 1e-13 * np.linalg.norm(yo)


def test_cpu_ref_protocol_golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_p2_2d_n32.npz"))
    od = o.uniform_mesh_desc(2, 2, 32)
    ref = cpu_ref.CpuRef(od, cpu_ref.structured_cell_colors([32, 32]))
    for k in (1, 2, 3):
        y = ref.bmop(k)
        # chained applies of the un-normalised operator amplify rounding differences by ||A|| ~ 100 per apply
        assert np.linalg.norm(y - g[f"prot{k}"]) <= 1e-12 * 100 ** (k - 1) * np.linalg.norm(g[f"prot{k}"])


def test_bmop_cpu_driver_runs_baseline_config_1():
    """BASELINE.json configs[0] as written: `bmop-cpu` built with DEGREE_FE=2, DIMENSION=2, 5 global refinements --
    bmop-cpu.cc's command line (max_refinement [min_refinement]) and its TSV line dim, degree, n_dofs, s per vmult."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "bin", "bmop-cpu-2d-p2")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle")])
    out = subprocess.run([exe, "5", "4"], stdout=subprocess.PIPE, text=True, timeout=300, check=True).stdout.strip().split("\n")
    assert len(out) == 2
    for line, ndofs in zip(out, (33 * 33, 65 * 65)):
        f = line.split("\t")
        assert f[:3] == ["2", "2", str(ndofs)] and float(f[3]) > 0.0
