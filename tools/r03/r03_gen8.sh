#!/bin/bash
OUT=gpurun_out/r03_gen8; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for v in gen_nolit gen_lit gen_nolit gen_lit; do
  echo "== $v" | tee -a $OUT/general_ab_literal_pass0.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>/dev/null | tee -a $OUT/general_ab_literal_pass0.txt
done
