#!/bin/bash
# general blind rotation: workgroups walk the rows of a step in their own orders (gen_rot) / the same order (gen_cur)
OUT=gpurun_out/r03_gen17; mkdir -p $OUT
REDSEC_HIP_LIB=$PWD/variants/lib_gen_rot.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for v in gen_cur gen_rot gen_cur gen_rot; do
  echo "== $v" | tee -a $OUT/general_ab_row_rotation.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_row_rotation.txt
done
