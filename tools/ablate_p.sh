#!/bin/bash
# Ablation builds of apply_planes3 (diagnostic only): libmfgpu_abl<N>.so with MFGPU_ABL=N
#   bit 0: no gather of source values   bit 1: no scatter stores   bit 2: no coefficient loads
# usage: tools/ablate_p.sh build  (here)   |   tools/ablate_p.sh run  (on the GPU box)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/dealii-cuda_amd
if [ "$1" = build ]; then
  for N in 1 2 4 7; do
    hipcc -O3 -std=c++17 -fPIC -DMFGPU_ABL=$N --offload-arch=gfx950 -c csrc/mfgpu_kernels_p.hip -o build/abl_p$N.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o lib/libmfgpu_abl$N.so build/abl_p$N.o $(ls build/mfgpu_*.o | grep -v kernels_p.o) || exit 1
  done
else
  for N in 0 1 2 4 7; do
    L=$R/dealii-cuda_amd/lib/libmfgpu_abl$N.so; [ $N = 0 ] && L=$R/dealii-cuda_amd/lib/libmfgpu.so
    echo -n "ABL=$N  "; MFGPU_LIB=$L python3 $R/bench.py --steps 30 --warmup 3 --no-cpu --no-second-line 2>&1 | tail -1 | grep -o "avg_launch_us[^,]*"
  done
fi
