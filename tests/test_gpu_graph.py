"""hipGraph capture of the vmult launches (launch-bound small problems: BASELINE config C1, 2D p=2 with 4 225 dofs, is
two ~5 us kernels per apply).  mfgpu_vmult only enqueues work on the caller's stream -- no allocation, no
synchronisation, no host read-back -- so a caller can capture the 100-apply loop of bmop.cu:142-146 once and replay it.
Checked: the replayed graph gives the same vectors as eager launches, for a one-segment and a segmented cell loop (the
side stream joins the capture through its events)."""
import numpy as np
import pytest

import pymfgpu as mf
from util import oracle_desc_from_mesh
from oracle import mf_oracle as o

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("dim,p,n,segs", [(2, 2, 32, 0), (3, 4, 6, 0), (3, 4, 8, 3), (3, 2, 9, 2)])
def test_vmult_loop_is_graph_capturable(dim, p, n, segs):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mesh = mf.Mesh.uniform(dim, p, n)
    mesh.desc.cell_loop_segments = segs
    od = oracle_desc_from_mesh(mesh)
    op = mf.Operator(mesh.desc, mesh)
    N = mesh.n_dofs
    dev = torch.device("cuda", 0)
    a = torch.full((N,), 0.1, device=dev, dtype=torch.float64)
    b = torch.zeros(N, device=dev, dtype=torch.float64)
    K = 4

    def loop(stream):
        x, y = a, b
        for _ in range(K):
            op.vmult(y, x, stream)
            x, y = y, x
        return x

    # eager reference
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        res = loop(s.cuda_stream).clone()
    torch.cuda.synchronize()
    want = res.cpu().numpy()
    ref = o.bmop_protocol(od, K, init=0.1) if hasattr(o, "bmop_protocol") else None
    if ref is not None:
        assert np.linalg.norm(want - ref) <= 1e-12 * 100 ** (K - 1) * np.linalg.norm(ref)
    # capture once, replay twice
    a.fill_(0.1)
    b.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = loop(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        a.fill_(0.1)
        b.zero_()
        g.replay()
        torch.cuda.synchronize()
        got = out.cpu().numpy()  # (pencil kernels sum a batch's cells with LDS atomics: last-bit differences between launches)
        assert np.linalg.norm(got - want) <= 1e-13 * np.linalg.norm(want)
