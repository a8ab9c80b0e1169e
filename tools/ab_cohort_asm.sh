#!/bin/bash
# Same-box A/B of the XCD cohort step written as one asm block (variants/lib_asmcohort.so) against the C++ form (lib_lagm1.so),
# all four lock-step kernel / parameter-set combinations, two alternating rounds; then the parity tests on the new form.
set -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
for ARGS in "--mode split --params redsec_small_v2" "--mode split" "--params redsec_small_v2" ""; do
  tools/ab_bench.sh 2 "--no-mnist --no-cifar --steps 3 $ARGS" lagm1 asmcohort
done 2>&1 | tee $OUT/ah_ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exactness.py -m gpu -x -q > $OUT/ah_tests.log 2>&1; rc=$?; tail -5 $OUT/ah_tests.log
exit $rc
