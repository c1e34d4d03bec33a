#!/usr/bin/env python3
"""Phase shares of apply_planes3 from the stamp build (make -C dealii-cuda_amd diag): s_memtime at the phase
boundaries of every batch (lane 0).  Read SHARES and cycles per batch, not run time."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MFGPU_LIB"] = os.path.join(ROOT, "dealii-cuda_amd", "lib", "libmfgpu_diag.so")
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import numpy as np  # noqa: E402

import pymfgpu as mf  # noqa: E402

adaptive = len(sys.argv) > 1 and sys.argv[1].startswith("adaptive:")  # adaptive:NREF -> the bmop ADAPTIVE_GRID mesh
n = 54 if adaptive or len(sys.argv) < 2 else int(sys.argv[1])
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nv = int(sys.argv[3]) if len(sys.argv) > 3 else 3  # vmults before the stamps are read (the last one's are kept)
mesh = mf.Mesh.adaptive(3, p, int(sys.argv[1].split(":")[1])) if adaptive else mf.Mesh.uniform(3, p, n)
op = mf.Operator(mesh.desc, mesh)
assert op.kernel_name().startswith("apply_planes3"), op.kernel_name()
L = mf.lib()
L.mfgpu_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.mfgpu_debug_stamps(op._h, None, 0) == 0
nbt = op.plan_stats()["n_batches"]
dst, src = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
dst.fill(0.1)
for i in range(nv):
    dst.swap(src)
    op.vmult(dst, src)
mf.synchronize()
buf = np.zeros(2 * nbt * 16, dtype=np.uint64)
assert L.mfgpu_debug_stamps(op._h, buf.ctypes.data, nbt) == 0
S = buf[:nbt * 16].reshape(nbt, 16).astype(np.int64)
names = ["top", "loads issued", "stage A", "stage B", "stage C", "results -> regs", "next staged (wait)", "scatter issued"]
ok = (S[:, 0] != 0) & (S[:, 7] != 0)
S = S[ok]
tot = (S[:, 7] - S[:, 0])
print(f"{ok.sum()} batches; cycles per batch: mean {tot.mean():.0f}, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
for k in range(1, 8):
    d = S[:, k] - S[:, k - 1]
    print(f"  {names[k-1]:>20s} -> {names[k]:<20s} mean {d.mean():8.0f} cyc ({100 * d.mean() / tot.mean():5.1f} %)  p90 {np.percentile(d, 90):8.0f}")
if adaptive:  # by tenth of the batch order (plain batches first, then the batches of cells with a mask)
    idx = np.nonzero(ok)[0]
    for t in range(10):
        m = (idx >= nbt * t // 10) & (idx < nbt * (t + 1) // 10)
        d = np.diff(S[m][:, :8], axis=1).mean(axis=0)
        print(f"  batches {nbt * t // 10:6d}..: {tot[m].mean():7.0f} cyc = " + " | ".join(f"{x:5.0f}" for x in d))
rt = (S[:, 9] - S[:, 8])
good = rt > 0
print(f"in-kernel clock (cycles / 100 MHz ticks): {100e6 * tot[good].sum() / rt[good].sum() / 1e9:.3f} GHz;"
      f" batch time {rt[good].mean() * 10:.0f} ns")

# per-workgroup timeline (100 MHz ticks): start of first batch, end of last, busy time
wg = S[:, 10] - 1
t0 = S[:, 8].min()
import collections
first, last, busy, cnt = {}, {}, collections.Counter(), collections.Counter()
for w, a, e in zip(wg, S[:, 8] - t0, S[:, 9] - t0):
    first[w] = min(first.get(w, 1 << 60), a)
    last[w] = max(last.get(w, 0), e)
    busy[w] += e - a
    cnt[w] += 1
ws = sorted(first)
f = np.array([first[w] for w in ws]) / 100.0
l = np.array([last[w] for w in ws]) / 100.0
bz = np.array([busy[w] for w in ws]) / 100.0
c = np.array([cnt[w] for w in ws])
print(f"{len(ws)} workgroups; batches per workgroup min {c.min()} max {c.max()}")
print(f"first batch starts (us): min {f.min():.1f} p50 {np.percentile(f, 50):.1f} max {f.max():.1f}")
print(f"last batch ends   (us): min {l.min():.1f} p50 {np.percentile(l, 50):.1f} max {l.max():.1f}")
print(f"in-batch time per workgroup (us): min {bz.min():.1f} p50 {np.percentile(bz, 50):.1f} max {bz.max():.1f};"
      f" between batches p50 {np.percentile(l - f - bz, 50):.1f}")

# per XCD (workgroups = x mod 8 share one under the dispatcher's round-robin placement) and per CU-sized group
wsa = np.array(ws)
for x in range(8):
    m = (wsa % 8) == x
    print(f"  workgroups = {x} (mod 8): in-batch time (us) p10 {np.percentile(bz[m], 10):.1f} p50 {np.percentile(bz[m], 50):.1f} "
          f"p90 {np.percentile(bz[m], 90):.1f}; last batch ends p50 {np.percentile(l[m], 50):.1f} max {l[m].max():.1f}; batches {c[m].sum()}")
per = bz / c
print(f"time per batch by workgroup (us): p10 {np.percentile(per, 10):.2f} p50 {np.percentile(per, 50):.2f} p90 {np.percentile(per, 90):.2f} max {per.max():.2f}")
