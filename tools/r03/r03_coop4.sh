#!/bin/bash
# cooperative kernel with rotated rows: half the key row prefetched across the transform (coop_half, default) / the whole row (coop_whole)
OUT=gpurun_out/r03_coop4; mkdir -p $OUT
for r in 1 2 3 4; do for v in coop_half coop_whole; do
  echo "== $v" | tee -a $OUT/mnist_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -E "whole image|B 196" | tee -a $OUT/mnist_ab.txt
done; done
