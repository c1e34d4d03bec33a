#!/bin/bash
# usage: tools/pmc_pass.sh OUTDIR TAG "COUNTER ..." -- prof_run args
# One rocprofv3 --pmc pass (few counters: a pass that exceeds the hardware slots aborts and hangs),
# bounded by timeout.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1; TAG=$2; CTRS=$3; shift 3; shift
mkdir -p $OUT
timeout -k 5 150 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/$TAG -- python3 $R/tools/prof_run.py "$@" > $OUT/$TAG.log 2>&1
echo "$TAG rc=$?"
