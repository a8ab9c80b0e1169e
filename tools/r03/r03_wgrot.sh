#!/bin/bash
# lock-step kernel, even l (REDsec set): row pairs of a component walked in per-workgroup rotated order (wg_rot) / same order (wg_cur)
OUT=gpurun_out/r03_wgrot; mkdir -p $OUT
bash tools/ab_bench.sh 3 "--steps 2 --warmup 1 --params redsec_small_v2" wg_cur wg_rot 2>&1 | tee $OUT/ab_redsec.txt
