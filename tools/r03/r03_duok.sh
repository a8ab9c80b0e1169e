#!/bin/bash
OUT=gpurun_out/r03_duok; mkdir -p $OUT
for r in 1 2; do for v in duok9 duok6 duok3; do
  echo "== $v" | tee -a $OUT/mnist_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 1024|B 600" | tee -a $OUT/mnist_ab.txt
done; done
