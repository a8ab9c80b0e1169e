#!/bin/bash
# decimation-in-time inverse pair in the lock-step workgroup kernel: parity, then same-box A/B against the Gentleman-Sande pair
OUT=gpurun_out/r03_dit; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exactness.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
grep -q failed $OUT/tests.txt && exit 1
bash tools/ab_bench.sh 3 "--steps 3 --warmup 1" wg_gs wg_dit 2>&1 | tee $OUT/ab_dit_default128.txt
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --params redsec_small_v2" wg_gs wg_dit 2>&1 | tee $OUT/ab_dit_redsec.txt
