#!/bin/bash
# basebit-1 keyswitch through scalar loads (default) against the tiled LDS kernel (RS_NO_KS_BIT=1): parity, then rates
OUT=gpurun_out/r03_ksbit; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_general.py tests/test_gpu_ops_wrappers.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for r in 1 2; do
  echo "== tiled (RS_NO_KS_BIT=1)" | tee -a $OUT/rates.txt
  RS_NO_KS_BIT=1 timeout -k 10 300 python tools/general_rate.py redsec_small redsec_medium redsec_large 2>/dev/null | tee -a $OUT/rates.txt
  echo "== bit form" | tee -a $OUT/rates.txt
  timeout -k 10 300 python tools/general_rate.py redsec_small redsec_medium redsec_large 2>/dev/null | tee -a $OUT/rates.txt
done
