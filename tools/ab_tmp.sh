R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  for which in prev cur; do
    if [ $which = prev ]; then export MFGPU_LIB=$R/dealii-cuda_amd/lib/libmfgpu_prev.so; else unset MFGPU_LIB; fi
    echo "== $which" >> $R/gpurun_out/ab.log
    python $R/bench.py --no-cpu --no-second-line "$@" 2>/dev/null | tail -1 >> $R/gpurun_out/ab.log
  done
done
