#!/bin/bash
# cooperative (latency) kernel: per-lane twiddles kept in registers (coop_kept) against read from the LDS table (coop_table)
OUT=gpurun_out/r03_coop; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mnist.py tests/test_gpu_relu.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for r in 1 2 3; do for v in coop_table coop_kept; do
  echo "== $v" | tee -a $OUT/mnist_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 196" | tee -a $OUT/mnist_ab.txt
done; done
