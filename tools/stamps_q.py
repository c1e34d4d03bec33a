#!/usr/bin/env python3
"""Phase shares of apply_planes4 from a stamp build (diagnostic only; the product source carries no stamps): a patched
COPY of mfgpu_kernels_q.hip with s_memtime stamps at the stage boundaries is compiled with -DMFGPU_STAMPS into
lib/libmfgpu_stampq.so.   tools/stamps_q.py build  (here)   |   tools/stamps_q.py run [n] [p]  (on the GPU box)
Read SHARES and cycles per batch, not run time (a stamp drains the LDS queue)."""
import ctypes as C
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(R, "dealii-cuda_amd")
ANCHORS = [("  while (true) {\n", "    STAMP(0);\n    RSTAMP(8);\n    if (A.stamps && threadIdx.x == 0) A.stamps[(size_t)b * 16 + 10] = blockIdx.x + 1;\n"),
           ("    // ---- S1 (xy)", "    STAMP(1);\n"), ("    // ---- S2 (yz)", "    STAMP(2);\n"), ("    // ---- S3 (xy)", "    STAMP(3);\n"),
           ("    // ---- S4 (yz)", "    STAMP(4);\n"), ("    // ---- everything the NEXT batch", "    STAMP(5);\n"),
           ("    // ---- S5 (xy)", "    STAMP(6);\n"), ("    // ---- S6: scatter", "    STAMP(7);\n"),
           ("    if (!has_next) break;\n", "    STAMP(11);\n    RSTAMP(9);\n")]
NAMES = ["S0 stage src", "S1 gather plane, S_y S_x D_x", "S2 x-part", "S3 D_x^T, a -> T", "S4 y/z parts", "prefetch issue",
         "S5 S_x^T S_y^T, adds", "S6 scatter"]
KEYS = [0, 1, 2, 3, 4, 5, 6, 7, 11]


def build():
    s = open(os.path.join(PKG, "csrc", "mfgpu_kernels_q.hip")).read()
    for anchor, ins in ANCHORS:
        assert s.count(anchor) == 1, anchor
        s = s.replace(anchor, ins + anchor if not anchor.startswith("  while (true)") else anchor + ins)
    src = os.path.join(PKG, "build", "stampq.hip")
    open(src, "w").write(s)
    srcs = [os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc"))
            if f.endswith((".hip", ".cpp")) and f != "mfgpu_kernels_q.hip"] + [src]
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-DMFGPU_STAMPS", "--offload-arch=gfx950", "-I", os.path.join(PKG, "csrc"), "-shared", "-o",
           os.path.join(PKG, "lib", "libmfgpu_stampq.so")]
    for f in srcs:
        cmd += (["-x", "hip"] if f.endswith(".hip") else ["-x", "c++"]) + [f]
    subprocess.check_call(cmd + ["-L/opt/rocm/lib", "-lrccl"])


def run():
    os.environ["MFGPU_LIB"] = os.path.join(PKG, "lib", "libmfgpu_stampq.so")
    sys.path.insert(0, PKG)
    import numpy as np
    import pymfgpu as mf
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 54
    p = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    mesh = mf.Mesh.uniform(3, p, n)
    mesh.desc.kernel = mf.KERNEL_PLANES_2W
    op = mf.Operator(mesh.desc, mesh)
    assert op.kernel_name() == "apply_planes4"
    L = mf.lib()
    L.mfgpu_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert L.mfgpu_debug_stamps(op._h, None, 0) == 0
    nbt = op.plan_stats()["n_batches"]
    dst, src = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
    dst.fill(0.1)
    for i in range(3):
        dst.swap(src)
        op.vmult(dst, src)
    mf.synchronize()
    buf = np.zeros(2 * nbt * 16, dtype=np.uint64)
    assert L.mfgpu_debug_stamps(op._h, buf.ctypes.data, nbt) == 0
    S = buf[:nbt * 16].reshape(nbt, 16).astype(np.int64)
    S = S[(S[:, 0] != 0) & (S[:, 11] != 0)]
    tot = S[:, 11] - S[:, 0]
    print(f"{len(S)} batches; cycles per batch (one wave): mean {tot.mean():.0f}, p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}")
    for i in range(len(KEYS) - 1):
        d = S[:, KEYS[i + 1]] - S[:, KEYS[i]]
        print(f"  {NAMES[i]:32s} mean {d.mean():8.0f} cyc ({100 * d.mean() / tot.mean():5.1f} %)  p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f}")
    rt = S[:, 9] - S[:, 8]
    good = rt > 0
    print(f"in-kernel clock: {100e6 * tot[good].sum() / rt[good].sum() / 1e9:.3f} GHz; batch time {rt[good].mean() * 10:.0f} ns")
    wg = S[:, 10] - 1
    t0 = S[:, 8].min()
    first, last = {}, {}
    for w, a, e in zip(wg, S[:, 8] - t0, S[:, 9] - t0):
        first[w] = min(first.get(w, 1 << 60), a)
        last[w] = max(last.get(w, 0), e)
    f = np.array([first[w] for w in sorted(first)]) / 100.0
    l = np.array([last[w] for w in sorted(first)]) / 100.0
    ws = np.array(sorted(first))
    for x in range(8):
        m = (ws % 8) == x
        print(f"  workgroups = {x} (mod 8): last batch ends (us) p10 {np.percentile(l[m], 10):.1f} p50 {np.percentile(l[m], 50):.1f} "
              f"p90 {np.percentile(l[m], 90):.1f} max {l[m].max():.1f}")
    print(f"{len(f)} workgroups; first batch starts (us): p50 {np.percentile(f, 50):.1f} max {f.max():.1f}; "
          f"last batch ends (us): min {l.min():.1f} p10 {np.percentile(l, 10):.1f} p50 {np.percentile(l, 50):.1f} p90 {np.percentile(l, 90):.1f} max {l.max():.1f}")


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
