"""Generates tests/golden/*.npz from the numpy oracle (oracle/mf_oracle.py).

The reference ships no golden vectors and cannot be run here (needs deal.II + nvcc), so these are
SELF-GENERATED: they pin the oracle against accidental change and give the GPU tests fixed inputs
and outputs; they are not reference outputs ("parity unpinned" beyond the reference's two test
procedures, see oracle/mf_oracle.py).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mf_oracle as o  # noqa: E402

CASES = {
    # name: (dim, p, n)      protocol fixtures of SURVEY.md 8c item 3
    "c1_p2_2d_n32": (2, 2, 32),
    "p4_3d_n2": (3, 4, 2),
    "p4_3d_n3": (3, 4, 3),
    "p6_3d_n2": (3, 6, 2),
    "p1_3d_n4": (3, 1, 4),
    "p3_2d_n5": (2, 3, 5),
}


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    for name, (dim, p, n) in CASES.items():
        d = o.uniform_mesh_desc(dim, p, n)
        rng = np.random.default_rng(1234)
        x = rng.standard_normal(d.n_dofs)
        y = o.vmult(d, x)
        prot = [o.bmop_protocol(d, k) for k in (1, 2, 3)]
        np.savez_compressed(os.path.join(out, name + ".npz"), dim=dim, degree=p, n=n, x=x, y=y,
                            prot1=prot[0], prot2=prot[1], prot3=prot[2])
        print(name, d.n_dofs, np.linalg.norm(y))


if __name__ == "__main__":
    main()
