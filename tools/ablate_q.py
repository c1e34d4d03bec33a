#!/usr/bin/env python3
"""Ablation builds of apply_planes4 (diagnostic only; the product source carries no switches): patched COPIES of
mfgpu_kernels_q.hip are compiled into lib/libmfgpu_ablq_<name>.so.
   tools/ablate_q.py build          (here, CPU)        tools/ablate_q.py run   (on the GPU box)
Variants (what is removed; values stay live through asm volatile so that nothing upstream is dead code):
   nogather  source values: SV = (T) of the dof id instead of a load      nostore   no scatter stores
   nocoef    coefficient rows: constant instead of loads                  norec     dof lists / index runs loaded once only
   nomem     all four                                                     nolds     nomem + transposes not through LDS
"""
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "dealii-cuda_amd", "csrc", "mfgpu_kernels_q.hip")
VARIANTS = {"x4c": ["x4c"], "x4ci": ["x4c", "x4i"], "nocoef": ["coef"], "nomem": ["gather", "store", "coef", "rec"],
            "nolds": ["gather", "store", "coef", "rec", "lds"]}

PATCH = {
    "gather": [("    for (int j = 0; j < KGU; ++j) SV[j] = src_at(Gn[j]);\n",
                "    for (int j = 0; j < KGU; ++j) SV[j] = (T)(Gn[j] & 1023u);\n")],
    "store": [("          *p = ADD ? *p + r : r;\n", "          asm volatile(\"\" ::\"v\"(r), \"v\"(p));\n"),
              ("          hp[(j - JI) * 64] = r;\n", "          asm volatile(\"\" ::\"v\"(r), \"v\"(hp));\n")],
    "coef": [("    load_coef(b1, Cc);\n",
              "    for (int r = 0; r < n2; ++r) asm volatile(\"\" : \"+v\"(Cc[r]));\n")],
    "rec": [("    load_dofs(b1, Gn);\n    load_ix(b1, IXn);\n",
             "    for (int j = 0; j < KGU; ++j) { Gn[j] = Gc[j]; asm volatile(\"\" : \"+v\"(Gn[j])); }\n"
             "    for (int w3 = 0; w3 < NIW; ++w3) { IXn[w3] = IXc[w3]; asm volatile(\"\" : \"+v\"(IXn[w3])); }\n")],
    # what 16-byte loads of the coalesced streams would buy: the SAME bytes (permuted within the batch's records: wrong
    # results, valid addresses) with 13 + 5 + 4 instead of 25 + 18 + 13 load instructions
    "x4c": [("    for (int r = 0; r < n2; ++r) c[r] = nt_load(p + r * NT);\n",
            "    for (int r = 0; r < (n2 + 1) / 2; ++r) {\n"
            "      const size_t o = (size_t)(2 * r * 64 + 2 * lane) < (size_t)(n2 * NT - 2) ? (size_t)(2 * r * 64 + 2 * lane) : (size_t)(n2 * NT - 2);\n"
            "      typedef double d2_t __attribute__((ext_vector_type(2)));\n      const d2_t v = nt_load(reinterpret_cast<const d2_t *>(A.coefp + (size_t)bb * (n2 * NT) + o));\n"
            "      c[2 * r] = (T)v.x;\n      if (2 * r + 1 < n2) c[2 * r + 1] = (T)v.y;\n    }\n")],
    "x4i": [
           ("    for (int w = 0; w < NIW; ++w) ix[w] = nt_load(p + w * NT);\n",
            "    for (int w = 0; w < (NIW + 3) / 4; ++w) {\n"
            "      const uint32_t o = (uint32_t)(w * 256 + 4 * lane) < (uint32_t)(NIW * NT - 4) ? (uint32_t)(w * 256 + 4 * lane) : (uint32_t)(NIW * NT - 4);\n"
            "      typedef unsigned u4_t __attribute__((ext_vector_type(4)));\n      const u4_t v = nt_load(reinterpret_cast<const u4_t *>(A.idxp + (size_t)bb * (NIW * NT) + o));\n"
            "      ix[4 * w] = v.x;\n      if (4 * w + 1 < NIW) ix[4 * w + 1] = v.y;\n      if (4 * w + 2 < NIW) ix[4 * w + 2] = v.z;\n      if (4 * w + 3 < NIW) ix[4 * w + 3] = v.w;\n    }\n")],
    # transposes: every Tw access becomes a register move kept alive (wrong results, same arithmetic)
    "lds": [(re.compile(r"(\w+(?:\[\w+\])+) = Tw\[[^;]*\];"), r'{ \1 = (T)lane; asm volatile("" : "+v"(\1)); }'),
            (re.compile(r"Tw\[[^;=]*\] = ([^;]*);"), r'asm volatile("" ::"v"(\1));'),
            (re.compile(r"lds_add\(&Tw\[[^;]*\], ([^;]*)\);"), r'asm volatile("" ::"v"(\1));')],
}


def patched(parts):
    s = open(SRC).read()
    for part in parts:
        for old, new in PATCH[part]:
            if isinstance(old, str):
                assert old in s, (part, old)
                s = s.replace(old, new)
            else:
                if False:
                    pass
                else:
                    s, k = old.subn(new, s)
                assert k > 0, (part, old.pattern)
    return s


def build():
    pkg = os.path.join(R, "dealii-cuda_amd")
    objs = [os.path.join(pkg, "build", f) for f in os.listdir(os.path.join(pkg, "build"))
            if f.startswith("mfgpu_") and f.endswith(".o") and f != "mfgpu_kernels_q.o"]
    for name, parts in VARIANTS.items():
        src = os.path.join(pkg, "build", f"ablq_{name}.hip")
        open(src, "w").write(patched(parts))
        obj = src[:-4] + ".o"
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(pkg, "csrc"),
                               "-c", src, "-o", obj])
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o",
                               os.path.join(pkg, "lib", f"libmfgpu_ablq_{name}.so"), obj] + objs + ["-L/opt/rocm/lib", "-lrccl"])
        print("built", name)


def run():
    for name in ["full"] + list(VARIANTS):
        lib = os.path.join(R, "dealii-cuda_amd", "lib", "libmfgpu.so" if name == "full" else f"libmfgpu_ablq_{name}.so")
        env = dict(os.environ, MFGPU_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--steps", "30", "--warmup", "3", "--no-cpu",
                              "--no-second-line"] + sys.argv[2:], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        m = re.search(r'"avg_launch_us": ([0-9.]+)', out)
        print(f"{name:10s} avg_launch_us {m.group(1) if m else '??'}", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
