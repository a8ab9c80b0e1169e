#!/bin/bash
OUT=gpurun_out/r03_duos; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -5 | tee $OUT/tests.txt || exit 1
REDSEC_MODE=split timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mnist_split_duo.txt
RS_NO_DUO=1 REDSEC_MODE=split timeout -k 10 200 python tools/mnist_latency.py 2>&1 | grep -v amdgpu.ids | tee $OUT/mnist_split_noduo.txt
