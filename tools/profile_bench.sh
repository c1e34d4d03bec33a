#!/bin/bash
# Round profile of the default bench command (run on the GPU box via gpurun):
#   kernel trace + stats, and separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE cannot
#   share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).  Summaries are copied to profiles/.
# usage: tools/profile_bench.sh TAG [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
# kernel durations in the state the timed region runs in (bench.py's clock ramp: 2000 untimed applies first); the PMC
# passes count bytes, which do not depend on the clocks, and skip the ramp (every dispatch is serialised under --pmc)
B="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-second-line --ramp-steps 0 $@"
BK="python3 $R/bench.py --steps 100 --warmup 10 --no-cpu --no-second-line $@"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BK > $O/kt.log 2>&1; echo "kt rc=$?"
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1; echo "write rc=$?"
timeout -k 5 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -- $B > $O/l2.log 2>&1; echo "l2 rc=$?"
# cross-check of the calibrated read side: the L2's fabric read requests by size (32 / 64 / 128 B)
timeout -k 5 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rdreq -- $B > $O/rdreq.log 2>&1; echo "rdreq rc=$?"
python3 $R/tools/profile_summary.py $O $R/gpurun_out/${TAG}_summary
