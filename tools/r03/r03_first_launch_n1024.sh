#!/bin/bash
# first launches of the N = 1024 kernels: fresh processes, each a short differential stress of the three modes (every kernel form's
# size class is among the first 15 rounds); a race that needs the desynchronised wavefronts of a first launch would show as a mismatch
OUT=gpurun_out/r03_first; mkdir -p $OUT
bad=0
for i in $(seq 1 16); do
  timeout -k 10 200 python tools/stress_modes.py 15 $((100 + i)) 2>&1 | grep -v amdgpu.ids | grep "done:" | tee -a $OUT/first_launch_n1024.txt | grep -v "mismatches 0" && bad=$((bad+1))
done
echo "fresh processes with a mismatch: $bad of 16" | tee -a $OUT/first_launch_n1024.txt
