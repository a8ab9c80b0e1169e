#!/bin/bash
OUT=gpurun_out/r03_race; mkdir -p $OUT
for v in gen_racy gen_fixed; do
  echo "== $v" | tee -a $OUT/race_repro.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 400 python tools/r03/r03_race_repro.py redsec_medium 1024 14 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $OUT/race_repro.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 400 python tools/r03/r03_race_repro.py redsec_large 512 6 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $OUT/race_repro.txt
done
REDSEC_HIP_LIB=$PWD/variants/lib_gen_fixed.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -3 | tee -a $OUT/race_repro.txt
