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
VARIANTS = {"stagger": ["stagger"], "stagger2": ["stagger2"], "nomem": ["gather", "store", "coef", "rec"], "nolds": ["gather", "store", "coef", "rec", "lds"],
            "valu": ["gather", "store", "coef", "rec", "lds", "ua"]}

PATCH = {
    "gather": [("        if (i < KGU) SV[i] = src_at(Gn[i]);\n", "        if (i < KGU) SV[i] = (T)(Gn[i] & 1023u);\n")],
    "store": [("          *p = ADD ? *p + r : r;\n", "          asm volatile(\"\" ::\"v\"(r), \"v\"(p));\n"),
              ("          hp[(j - JI) * 64] = r;\n", "          asm volatile(\"\" ::\"v\"(r), \"v\"(hp));\n")],
    "coef": [("        else if (!kCoefInS1) Cc[i - KGU - NIW] = nt_load(cnext + (i - KGU - NIW) * NT);\n",
              "        else if (!kCoefInS1) asm volatile(\"\" : \"+v\"(Cc[i - KGU - NIW]));\n"),
             ("      for (int r = (n2 * s) / (2 * n); r < (n2 * (s + 1)) / (2 * n); ++r) Cc[r] = nt_load(cthis + r * NT);\n",
              "      for (int r = (n2 * s) / (2 * n); r < (n2 * (s + 1)) / (2 * n); ++r) asm volatile(\"\" : \"+v\"(Cc[r]));\n")],
    "rec": [("        else if (i < KGU + NIW) IXn[i - KGU] = nt_load(ixnext + (i - KGU) * NT);\n",
             "        else if (i < KGU + NIW) { IXn[i - KGU] = IXc[i - KGU]; asm volatile(\"\" : \"+v\"(IXn[i - KGU])); }\n"),
            ("    uint32_t Gn[KGU];\n    load_dofs(b1, Gn);\n",
             "    uint32_t Gn[KGU];\n    for (int j = 0; j < KGU; ++j) { Gn[j] = Gc[j]; asm volatile(\"\" : \"+v\"(Gn[j])); }\n")],
    # experiment: the second half of the grid (the second wave of every SIMD) starts half a batch later
    "stagger": [("  if (b >= bend) return;\n", "  if (b >= bend) return;\n  if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(110);\n")],
    "stagger2": [("  if (b >= bend) return;\n", "  if (b >= bend) return;\n  if (blockIdx.x & 1) { __builtin_amdgcn_s_sleep(110); }\n")],
    # transposes: every Tw access becomes a register move kept alive (wrong results, same arithmetic)
    "lds": [(re.compile(r"(\w+(?:\[\w+\])+) = Tw\[[^;]*\];"), r'{ \1 = (T)lane; asm volatile("" : "+v"(\1)); }'),
            (re.compile(r"Tw\[[^;=]*\] = ([^;]*);"), r'asm volatile("" ::"v"(\1));'),
            (re.compile(r"lds_add\(&Tw\[[^;]*\], ([^;]*)\);"), r'asm volatile("" ::"v"(\1));')],
    # ... and the batch array (gathered values -> planes, zero, accumulate, results): arithmetic only is left
    "ua": [("      ua[lane + j * 64] = v;\n", "      asm volatile(\"\" ::\"v\"(v));\n"),
           ("      u[i] = (T) * reinterpret_cast<const double *>(reinterpret_cast<const char *>(ua) + ixb(IXc, i));\n",
            "      { u[i] = (T)(lane + i + ixb(IXc, i)); asm volatile(\"\" : \"+v\"(u[i])); }\n"),
           ("    for (int j = 0; j < (NUA + 63) / 64; ++j) ua[lane + j * 64] = 0.0;\n", "    for (int j = 0; j < 1; ++j) {}\n"),
           ("      lds_add(reinterpret_cast<double *>(reinterpret_cast<char *>(ua) + ixb(IXc, i)), (double)w[i]);\n",
            "      asm volatile(\"\" ::\"v\"(w[i]), \"v\"(ixb(IXc, i)));\n"),
           ("        const T r = (T)ua[lane + j * 64];\n", "        T r = (T)(lane + j); asm volatile(\"\" : \"+v\"(r));\n")],
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
                              "--no-second-line", "--kernel", "planes_2w"] + sys.argv[2:], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        m = re.search(r'"avg_launch_us": ([0-9.]+)', out)
        print(f"{name:10s} avg_launch_us {m.group(1) if m else '??'}", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
