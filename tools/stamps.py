#!/usr/bin/env python3
"""Phase-share diagnosis with the stamp build (make -C dealii-cuda_amd diag).  Reads SHARES, never
run time: the stamped build forbids overlaps the product kernel has."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MFGPU_LIB"] = os.path.join(ROOT, "dealii-cuda_amd", "lib", "libmfgpu_diag.so")
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import numpy as np  # noqa: E402

import pymfgpu as mf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
bc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mesh = mf.Mesh.uniform(3, p, n)
mesh.desc.max_cells_per_batch = bc
op = mf.Operator(mesh.desc, mesh)
L = mf.lib()
L.mfgpu_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.mfgpu_debug_stamps(op._h, None, 0) == 0
st = op.plan_stats()
nbt = st["n_batches"]
dst, src = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
dst.fill(0.1)
for i in range(3):
    dst.swap(src)
    op.vmult(dst, src)
mf.synchronize()
buf = np.zeros(2 * nbt * 16, dtype=np.uint64)
assert L.mfgpu_debug_stamps(op._h, buf.ctypes.data, nbt) == 0
S = buf[:nbt * 16].reshape(nbt, 16).astype(np.int64)
PS = buf[nbt * 16:].reshape(nbt, 16).astype(np.int64)
print("plan", st)
plane = os.environ.get("MFGPU_PLANE") == "1"
names = {0: "entry", 1: "loads issued", 2: "gathered->LDS (wait SV, lmap)", 3: "stage 1 (A: gather, S_x, S_y)",
         4: "stage 2 (B: S_z, D_z, D_x; wait cB)", 5: "stage 3 (A: D_y, S^T; wait cA)", 6: "stage 4 (B: S_z^T, acc)",
         7: "scatter issued"} if plane else {0: "entry", 2: "src gathered", 3: "gather barrier", 4: "chunk0 done", 5: "chunk1 done",
         6: "chunk2 done", 7: "chunk3 done", 13: "all chunks+barrier", 15: "scatter drained"}
used = [k for k in sorted(names) if (S[:, k] != 0).any()]
prev = used[0]
tot = (S[:, used[-1]] - S[:, used[0]])
print(f"workgroup lifetime: mean {tot.mean():.0f} cyc, p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f}")
for k in used[1:]:
    d = S[:, k] - S[:, prev]
    print(f"  {names[prev]:>24s} -> {names[k]:<24s} mean {d.mean():8.0f} cyc  ({100*d.mean()/tot.mean():5.1f} %)  p90 {np.percentile(d,90):8.0f}")
    prev = k
# dispatch ramp of the first colour
nb0 = st["n_batches"] // st["n_launches"]
e = S[:nb0, 0] - S[:nb0, 0].min()
x = S[:nb0, 15] - S[:nb0, 0].min()
print(f"colour 0: {nb0} workgroups; start offsets p50 {np.percentile(e,50):.0f} p90 {np.percentile(e,90):.0f} max {e.max():.0f};"
      f" end max {x.max():.0f} cyc")

# per-stage stamps of the SECOND chunk of every batch (3D): P0 .. P6
ok = (PS[:, 0] != 0) & (PS[:, 7] != 0)
if ok.any():
    names = ["P0 gather+S_x", "P1 S_y", "P2 S_z,D_z", "P3 D_y", "P4 D_x,S_x^T", "P5 S_y^T+stage", "P6 S_z^T+acc"]
    tot = (PS[ok, 7] - PS[ok, 0]).mean()
    print(f"second chunk, stage shares (wave 0, {ok.sum()} batches), total {tot:.0f} cyc:")
    for k in range(7):
        d = PS[ok, k + 1] - PS[ok, k]
        print(f"  {names[k]:>16s} mean {d.mean():7.0f} cyc ({100*d.mean()/tot:4.1f} %)  p90 {np.percentile(d,90):7.0f}")
