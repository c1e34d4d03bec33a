#!/bin/bash
# Bench line + rocprofv3 kernel stats of the secondary configurations (run on the GPU box via gpurun):
#   tools/profile_configs.sh TAG      -> gpurun_out/TAG_<cfg>_line.json, gpurun_out/TAG_<cfg>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
run() {  # name, bench args...
  local name=$1; shift
  python3 $R/bench.py --steps 50 --warmup 5 --no-cpu --no-second-line "$@" 2>/dev/null | tail -1 > $R/gpurun_out/${TAG}_${name}_line.json
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-second-line "$@" > $R/gpurun_out/${TAG}_${name}_kt.log 2>&1
  cp $(ls $R/gpurun_out/${TAG}_${name}_kt/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_${name}_kernel_stats.csv
  echo "$name done: $(cut -c1-200 $R/gpurun_out/${TAG}_${name}_line.json | grep -o '"ms_per_step": [0-9.]*')"
}
run c3 --adaptive 6 && run c5 --degree 6 --cells 36 && run float --float && run ball --ball 5 && run general --general-jacobian 0.1
run n64 --cells 64 && run n108 --cells 108 && run p5 --degree 5 --cells 43 && run planes2w --kernel planes_2w
