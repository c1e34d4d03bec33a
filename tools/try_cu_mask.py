#!/usr/bin/env python3
"""What a CU-masked launch stream costs the cell loop (hipExtStreamCreateWithCUMask): bench.py --gpus N > 1 launches on a
stream that leaves a few CUs to RCCL's send / recv kernel, which (19.7 KB of LDS, 261-280 registers per lane, four waves
per workgroup: llvm-readelf --notes of librccl's gfx950 code object) cannot share a CU with the plane kernel's four
one-wave workgroups.  Also shows how mask bits map to XCDs (bit i -> XCD i mod 8 if the first variant is the cheap one).
usage: tools/try_cu_mask.py [cells] [K]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dealii-cuda_amd"))
import torch  # noqa: E402,F401  (its libamdhip64 is the process's HIP runtime)

import pymfgpu as mf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
hip = C.CDLL("libamdhip64.so.7") if not hasattr(C, "_hip") else None
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def masked_stream(off_bits):
    words = (C.c_uint32 * 8)(*([0xffffffff] * 8))
    for b in off_bits:
        words[b // 32] &= ~(1 << (b % 32)) & 0xffffffff
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


def run(stream, max_wg):
    mesh = mf.Mesh.uniform(3, 4, n)
    mesh.desc.max_workgroups = max_wg
    op = mf.Operator(mesh.desc, mesh)
    a, b = mf.DeviceVector(mesh.n_dofs), mf.DeviceVector(mesh.n_dofs)
    a.fill(0.1)
    sp = stream.value if stream is not None else None
    out = []
    for rep in range(3):
        a.fill(0.1)
        mf.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            a, b = b, a
            op.vmult(a, b, sp)
        mf.synchronize()
        out.append((time.perf_counter() - t0) / K * 1e3)
    return min(out)


print(f"no mask, 1024 workgroups:                      {run(None, 0):.4f} ms per vmult")
print(f"no mask, 992 workgroups:                       {run(None, 992):.4f}")
s1 = masked_stream(range(8))
print(f"bits 0..7 off (one CU per XCD?), 992:          {run(s1, 992):.4f}")
s2 = masked_stream(range(0, 64, 8))
print(f"bits 0, 8, .., 56 off (eight CUs of XCD 0?), 992: {run(s2, 992):.4f}")
s3 = masked_stream(range(16))
print(f"bits 0..15 off (two CUs per XCD?), 960:        {run(s3, 960):.4f}")
