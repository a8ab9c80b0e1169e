#!/bin/bash
# split cooperative kernel with rotated rows: key chunks requested across the transform: 1 (default so far), 2, 4
OUT=gpurun_out/r03_coop6; mkdir -p $OUT
for r in 1 2 3; do for v in coops_a1 coops_a2 coops_a4; do
  echo "== $v" | tee -a $OUT/mnist_split_ab.txt
  REDSEC_MODE=split REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 196" | tee -a $OUT/mnist_split_ab.txt
done; done
REDSEC_HIP_LIB=$PWD/variants/lib_coops_a4.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py tests/test_gpu_mnist.py -x -q 2>&1 | tail -2 | tee -a $OUT/mnist_split_ab.txt
