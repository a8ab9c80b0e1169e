#!/bin/bash
OUT=gpurun_out/r03_gpu3; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=30 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
tail -n 60 $OUT/pytest_gpu.log
true
