#!/bin/bash
# general blind rotation with per-ring row rotation as the default: parity, then lookahead variants again (the balance may have moved)
OUT=gpurun_out/r03_gen19; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for v in gen_cur gen_fg2 gen_la1 gen_cur gen_fg2 gen_la1; do
  echo "== $v" | tee -a $OUT/general_ab_after_rotation.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_after_rotation.txt
done
