#!/bin/bash
OUT=gpurun_out/r03_wgsk; mkdir -p $OUT
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --mode split" wgsk9 wgsk6 wgsk3 2>&1 | tee $OUT/ab_split_default128.txt
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --mode split --params redsec_small_v2" wgsk9 wgsk6 wgsk3 2>&1 | tee $OUT/ab_split_redsec.txt
