#!/bin/bash
# rocprofv3 --pmc passes (each in its own run, counters only) of one 65,536-gate bench step, both parameter
# sets -> gpurun_out/pmc/<params>_<set>.csv ; folded afterwards by tools/pmc_traffic.py (FETCH/WRITE).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc; mkdir -p $OUT
for P in default128 redsec_small_v2; do
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    TAG=$(echo $SET | cut -d' ' -f1)
    rm -rf $OUT/tmp
    rocprofv3 --pmc $SET --output-format csv -d $OUT/tmp -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-exact-check --params $P > $OUT/${P}_${TAG}.log 2>&1
    f=$(find $OUT/tmp -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then grep -E "Counter_Name|blind_rotate|keyswitch" "$f" > $OUT/${P}_${TAG}.csv; echo "$P $TAG: $(wc -l < $OUT/${P}_${TAG}.csv) rows"; else echo "$P $TAG: no counter file"; tail -3 $OUT/${P}_${TAG}.log; fi
  done
done
rm -rf $OUT/tmp
