#!/bin/bash
# cooperative kernel: rows of a step walked in a per-workgroup rotated order (coop_rot1), and the row groups rotated over the waves too (coop_rot2)
OUT=gpurun_out/r03_coop2; mkdir -p $OUT; rm -f $OUT/mnist_ab2.txt
for r in 1 2 3 4; do for v in coop_cur coop_rot1 coop_rot2; do
  echo "== $v" | tee -a $OUT/mnist_ab2.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 196" | tee -a $OUT/mnist_ab2.txt
done; done
REDSEC_HIP_LIB=$PWD/variants/lib_coop_rot2.so timeout -k 10 400 python -m pytest tests/test_gpu_mnist.py tests/test_gpu_parity.py tests/test_gpu_relu.py -x -q 2>&1 | tail -2 | tee -a $OUT/mnist_ab2.txt
