#!/bin/bash
# same-box A/B of the general blind rotation: key lookahead depth (positions requested ahead), SGPR-base key addressing
OUT=gpurun_out/r03_gen2; mkdir -p $OUT
REDSEC_HIP_LIB=$PWD/variants/lib_gen_la2.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -3 | tee -a $OUT/general_ab_lookahead.txt || exit 1
for v in gen_base gen_la1 gen_la2 gen_la3 gen_base gen_la1 gen_la2 gen_la3; do
  echo "== $v" | tee -a $OUT/general_ab_lookahead.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>/dev/null | tee -a $OUT/general_ab_lookahead.txt || exit 1
done
