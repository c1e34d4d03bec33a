"""Oracle against the committed golden fixtures (CPU) -- pins the oracle against drift."""
import glob
import os

import numpy as np
import pytest

from oracle import mf_oracle as o

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    d = o.uniform_mesh_desc(int(g["dim"]), int(g["degree"]), int(g["n"]))
    y = o.vmult(d, g["x"])
    assert np.linalg.norm(y - g["y"]) <= 1e-13 * np.linalg.norm(g["y"])
    p3 = o.bmop_protocol(d, 3)
    assert np.linalg.norm(p3 - g["prot3"]) <= 1e-12 * np.linalg.norm(g["prot3"])
