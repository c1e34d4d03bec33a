#!/bin/bash
# A/B of bench.py variants on one box: tools/ab.sh REPS "args A" "args B" ...   (kernel / pass-2 / step time per run)
R=${GRAFT_REPO_ROOT:-/root/repo}
REPS=$1; shift
for r in $(seq $REPS); do
  for a in "$@"; do
    echo -n "[$a] "
    python3 $R/bench.py --steps 50 --warmup 5 --no-cpu --no-second-line $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step %.4f kernel %.4f pass2 %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_vmult'], d['roofline']['pass2_ms_per_vmult']))"
  done
done
