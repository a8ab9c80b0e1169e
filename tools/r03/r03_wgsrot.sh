#!/bin/bash
OUT=gpurun_out/r03_wgsrot; mkdir -p $OUT
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --mode split" wgs_cur wgs_rot 2>&1 | tee $OUT/ab_split.txt
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --mode split --params redsec_small_v2" wgs_cur wgs_rot 2>&1 | tee -a $OUT/ab_split.txt
