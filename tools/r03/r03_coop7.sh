#!/bin/bash
# cooperative kernel: one row (coop_one) / two rows (coop_two) of key in flight per wave
OUT=gpurun_out/r03_coop7; mkdir -p $OUT
for r in 1 2 3; do for v in coop_one coop_two; do
  echo "== $v" | tee -a $OUT/mnist_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 196" | tee -a $OUT/mnist_ab.txt
done; done
REDSEC_HIP_LIB=$PWD/variants/lib_coop_two.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mnist.py tests/test_gpu_exactness.py -x -q 2>&1 | tail -2 | tee -a $OUT/mnist_ab.txt
