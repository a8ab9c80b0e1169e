#!/bin/bash
# general blind rotation with the last pass's twiddles kept: parity of the default build, then N = 8192 with its near levels staged in LDS too (gen_tw13)
OUT=gpurun_out/r03_gen15; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
REDSEC_HIP_LIB=$PWD/variants/lib_gen_tw13.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q -k "large or 8192" 2>&1 | tail -3 | tee -a $OUT/tests.txt
for v in gen_cur gen_tw13 gen_cur gen_tw13; do
  echo "== $v" | tee -a $OUT/general_ab_large_twiddles_in_lds_again.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_large_twiddles_in_lds_again.txt
done
