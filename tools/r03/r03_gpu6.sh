#!/bin/bash
OUT=gpurun_out/r03_gpu6; mkdir -p $OUT
free -g | tee $OUT/mem.txt; nproc | tee -a $OUT/mem.txt
for v in genbar genwave genbar genwave; do
  echo "== $v" | tee -a $OUT/general_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>/dev/null | tee -a $OUT/general_ab.txt
done
timeout -k 10 400 python -m pytest tests/test_gpu_general.py -m gpu -q --durations=8 > $OUT/pytest_general.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_general.log; tail -n 20 $OUT/pytest_general.log
