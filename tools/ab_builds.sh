#!/bin/bash
# A/B of two BUILDS of the library on one GPU box (boxes differ by +-2 %, runs on one box by +-0.5 %): three interleaved
# runs of bench.py with lib/libmfgpu_prev.so (build it from the sources to compare against, then copy it there) and with
# lib/libmfgpu.so.  Run on the GPU box: gpurun -- 'bash tools/ab_builds.sh [bench args]'; prints ms per vmult, cell loop
# and pass 2 (ms) per run.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_builds.log
rm -f $OUT
for i in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then export MFGPU_LIB=$R/dealii-cuda_amd/lib/libmfgpu_prev.so; else unset MFGPU_LIB; fi
    echo "== $which" >> $OUT
    python3 $R/bench.py --no-cpu --no-second-line "$@" 2>/dev/null | tail -1 >> $OUT
  done
done
python3 - $OUT <<'PY'
import json, sys
w = None
for l in open(sys.argv[1]):
    if l.startswith("=="):
        w = l.strip()
    elif l.startswith("{"):
        j = json.loads(l)
        r = j["roofline"]
        print(w, round(j["ms_per_step"], 4), round(r["kernel_ms_per_vmult"], 4), round(r["pass2_ms_per_vmult"], 4))
PY
