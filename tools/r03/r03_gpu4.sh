#!/bin/bash
OUT=gpurun_out/r03_gpu4; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_relu.py tests/test_gpu_ops_wrappers.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -n 30 $OUT/pytest.log
