#!/bin/bash
# twiddles fetched once per pipelined pair (tw_share) against once per transform (tw_noshare): parity, then same-box A/B
OUT=gpurun_out/r03_share; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exactness.py tests/test_gpu_mnist.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
grep -q failed $OUT/tests.txt && exit 1
bash tools/ab_bench.sh 3 "--steps 3 --warmup 1" tw_noshare tw_share 2>&1 | tee $OUT/ab_default128.txt
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --params redsec_small_v2" tw_noshare tw_share 2>&1 | tee $OUT/ab_redsec.txt
bash tools/ab_bench.sh 2 "--steps 2 --warmup 1 --mode split" tw_noshare tw_share 2>&1 | tee $OUT/ab_split.txt
