#!/bin/bash
# Ablation builds of apply_planes3 (diagnostic only): patched COPIES of mfgpu_kernels_p.hip (the product source carries
# no switches) -> lib/libmfgpu_abl<N>.so, N = bits  1: no gather of source values  2: no scatter stores  4: no
# coefficient loads.   usage: tools/ablate_p.sh build  (here)   |   tools/ablate_p.sh run  (on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/dealii-cuda_amd
if [ "$1" = build ]; then
  for N in 1 2 4 7; do
    python3 - $N <<'PY' || exit 1
import sys
n = int(sys.argv[1])
s = open("csrc/mfgpu_kernels_p.hip").read()
def rep(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)
if n & 1: rep("        SVn[j] = src_at(Gn[j]);\n", "        SVn[j] = (T)Gn[j];\n")
if n & 2: rep("        scatter_slot(j, bp, Gp[j], R[j], old[j]);\n", '        asm volatile("" ::"v"(R[j]), "v"(Gp[j]));\n')
if n & 4: rep("        Cc[r] = nt_load(cnext + r * NT);\n", '        asm volatile("" : "+v"(Cc[r]) : "v"(cnext));\n')
open("build/abl_p%d.hip" % n, "w").write(s)
PY
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I csrc -c build/abl_p$N.hip -o build/abl_p$N.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libmfgpu_abl$N.so build/abl_p$N.o $(ls build/mfgpu_*.o | grep -v kernels_p.o) -L/opt/rocm/lib -lrccl || exit 1
  done
else
  for N in 0 1 2 4 7; do
    L=$R/dealii-cuda_amd/lib/libmfgpu_abl$N.so; [ $N = 0 ] && L=$R/dealii-cuda_amd/lib/libmfgpu.so
    echo -n "ABL=$N  "; MFGPU_LIB=$L python3 $R/bench.py --steps 30 --warmup 3 --no-cpu --no-second-line 2>&1 | tail -1 | grep -o "avg_launch_us[^,]*"
  done
fi
