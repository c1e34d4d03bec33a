#!/bin/bash
# Counter passes of the plane kernel on the headline workload (run on the GPU box via gpurun):
#   tools/pmc_q.sh TAG [prof_run args]   -> gpurun_out/TAG/{kt,sq1,sq2,sq3}.  SQ counters: 8 per pass.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-q}; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
P="python3 $R/tools/prof_run.py $@"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $P > $O/kt.log 2>&1; echo "kt rc=$?"
timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq1 -- $P > $O/sq1.log 2>&1; echo "sq1 rc=$?"
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -- $P > $O/sq2.log 2>&1; echo "sq2 rc=$?"
timeout -k 5 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/sq3 -- $P > $O/sq3.log 2>&1; echo "sq3 rc=$?"
timeout -k 5 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/ta -- $P > $O/ta.log 2>&1; echo "ta rc=$?"
python3 $R/tools/pmc_summary.py $O ${PMC_KERNEL:-apply_planes} > $O/summary.txt 2>&1
python3 - <<PY >> $O/summary.txt
import csv,glob
for f in glob.glob("$O/kt/*/*kernel_trace.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "apply_planes" in r["Kernel_Name"]]
    if rows:
        r=rows[-1]
        d=[(int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3 for x in rows]
        print("kernel_trace:", {k:r[k] for k in r if k in ("Grid_Size","Workgroup_Size","LDS_Block_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Scratch_Size")}, "launches", len(d), "avg_us", sum(d)/len(d), "min", min(d), "max", max(d))
PY
cat $O/summary.txt
