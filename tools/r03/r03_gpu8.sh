#!/bin/bash
OUT=gpurun_out/r03_gpu8; mkdir -p $OUT
for v in gentw genpipe gentw genpipe; do
  echo "== $v" | tee -a $OUT/general_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>/dev/null | tee -a $OUT/general_ab.txt
done
timeout -k 10 500 python -m pytest tests/test_gpu_general.py -m gpu -q --durations=5 > $OUT/pytest_general.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_general.log; tail -n 12 $OUT/pytest_general.log
