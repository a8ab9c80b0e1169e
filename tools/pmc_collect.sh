#!/bin/bash
# rocprofv3 --pmc passes (each in its own run, counters only) of one 65,536-gate bench step ->
# gpurun_out/pmc/<params>_<mode>_<set>.csv ; FETCH/WRITE folded afterwards by tools/pmc_traffic.py.
# usage: tools/pmc_collect.sh [mode ...]      modes: fft (default) split exact
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc; mkdir -p $OUT
MODES="${@:-fft}"
for P in default128 redsec_small_v2; do
 for M in $MODES; do
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    TAG=$(echo $SET | cut -d' ' -f1)
    rm -rf $OUT/tmp
    rocprofv3 --pmc $SET --output-format csv -d $OUT/tmp -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-exact-check --no-mnist --no-cifar --no-live-traffic --params $P --mode $M > $OUT/${P}_${M}_${TAG}.log 2>&1
    f=$(find $OUT/tmp -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then grep -E "Counter_Name|blind_rotate|keyswitch" "$f" > $OUT/${P}_${M}_${TAG}.csv; echo "$P $M $TAG: $(wc -l < $OUT/${P}_${M}_${TAG}.csv) rows"; else echo "$P $M $TAG: no counter file"; tail -3 $OUT/${P}_${M}_${TAG}.log; fi
  done
 done
done
rm -rf $OUT/tmp
