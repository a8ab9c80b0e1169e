#!/bin/bash
# FFT duo kernel: row pairs of a step walked in per-workgroup rotated order (duo_rot) / same order everywhere (duo_norot)
OUT=gpurun_out/r03_duo3; mkdir -p $OUT
for r in 1 2 3 4; do for v in duo_norot duo_rot; do
  echo "== $v" | tee -a $OUT/mnist_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 1024|B 600" | tee -a $OUT/mnist_ab.txt
done; done
REDSEC_HIP_LIB=$PWD/variants/lib_duo_rot.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mnist.py -x -q 2>&1 | tail -2 | tee -a $OUT/mnist_ab.txt
