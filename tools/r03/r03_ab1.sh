#!/bin/bash
# same-box A/B of the fair-share priority variants (tools/build_variant.sh names) + stamps of the fair2 build
OUT=gpurun_out/r03_ab1; mkdir -p $OUT
tools/ab_bench.sh 3 "--steps 3 --warmup 1" base fair0 fair1 fair2 fair23 2>&1 | tee $OUT/ab_default128.txt
tools/ab_bench.sh 2 "--steps 2 --warmup 1 --params redsec_small_v2" base fair0 fair1 fair2 fair23 2>&1 | tee $OUT/ab_redsec.txt
REDSEC_HIP_LIB=$PWD/variants/lib_stamps_fair2.so timeout -k 10 300 python tools/stamp_profile.py default128 16384 > $OUT/stamps_fair2_default128.json 2> $OUT/stamps_fair2.err
cat $OUT/stamps_fair2_default128.json
