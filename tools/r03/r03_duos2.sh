#!/bin/bash
# split duo kernel: digits of a step walked in per-workgroup rotated order (duos_rot) / same order everywhere (duos_norot)
OUT=gpurun_out/r03_duos2; mkdir -p $OUT
for r in 1 2 3; do for v in duos_norot duos_rot; do
  echo "== $v" | tee -a $OUT/mnist_split_ab.txt
  REDSEC_MODE=split REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 1024|B 600" | tee -a $OUT/mnist_split_ab.txt
done; done
REDSEC_HIP_LIB=$PWD/variants/lib_duos_rot.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py tests/test_gpu_mnist.py -x -q 2>&1 | tail -2 | tee -a $OUT/mnist_split_ab.txt
