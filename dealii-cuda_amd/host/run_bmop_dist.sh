#!/bin/bash
# usage: run_bmop_dist.sh N_GPUS [cells per direction of the global mesh] [degree]
# Starts one bmop-dist process per GPU of this node (HIP_VISIBLE_DEVICES = rank) and waits for all of them.
N=${1:-1}; CELLS=${2:-54}; DEG=${3:-4}
HERE=$(cd "$(dirname "$0")" && pwd)
ID=$(mktemp -u /tmp/mfgpu_bmop_dist.XXXXXX)
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 0; r < N; r++)); do
  MFGPU_RANK=$r MFGPU_WORLD=$N MFGPU_ID_FILE=$ID HIP_VISIBLE_DEVICES=$r "$HERE/bin/bmop-dist" "$CELLS" "$DEG" &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
rm -f "$ID" "$ID".t*
exit $rc
