#!/bin/bash
# Run ON THE GPU BOX: PC sampling of the bench step (where do the waves of the count pass spend their time?).
export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r3
cd $R
rocprofv3 -L > gpurun_out/r3/pcs_list.txt 2>&1
grep -i -B2 -A12 "sampl" gpurun_out/r3/pcs_list.txt | head -60
for method in stochastic host_trap; do
  unit=cycles; interval=1048576
  if [ $method = host_trap ]; then unit=time; interval=1; fi
  rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $interval --kernel-trace -d gpurun_out/r3/pcs_$method -o pcs --output-format csv -- python3 bench.py --no-cpu-baseline --no-configs --extra-n 0 --steps 50 --warmup 5 > gpurun_out/r3/pcs_$method.log 2>&1
  tail -3 gpurun_out/r3/pcs_$method.log
  ls -la gpurun_out/r3/pcs_$method 2>/dev/null | head
done
