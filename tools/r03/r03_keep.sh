#!/bin/bash
# per-lane twiddles of the lock-step kernel kept in registers: none (keep9) / last group (keep6) / both groups (keep3)
OUT=gpurun_out/r03_keep; mkdir -p $OUT
bash tools/ab_bench.sh 2 "--steps 3 --warmup 1" keep9 keep6 keep3 2>&1 | tee $OUT/ab_default128.txt
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --params redsec_small_v2" keep9 keep6 keep3 2>&1 | tee $OUT/ab_redsec.txt
