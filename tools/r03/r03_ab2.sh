#!/bin/bash
# same-box A/B: planar exchange read back with 16-byte loads (wide, the new default) against eight 8-byte loads (b64)
OUT=gpurun_out/r03_ab2; mkdir -p $OUT
tools/ab_bench.sh 3 "--steps 3 --warmup 1" b64 wide 2>&1 | tee $OUT/ab_default128.txt
tools/ab_bench.sh 2 "--steps 2 --warmup 1 --params redsec_small_v2" b64 wide 2>&1 | tee $OUT/ab_redsec.txt
tools/ab_bench.sh 2 "--steps 2 --warmup 1 --mode split" b64 wide 2>&1 | tee $OUT/ab_split.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exactness.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_parity.txt
