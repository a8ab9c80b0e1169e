#!/bin/bash
# phase costs of the general blind rotation by timing-only variants + fabric-side traffic of one launch
OUT=gpurun_out/r03_gen1; mkdir -p $OUT
export TMPDIR=/tmp
for v in gen_base gen_nokey gen_nofwd gen_noinv gen_base; do
  echo "== $v" | tee -a $OUT/general_phase_probes.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>/dev/null | tee -a $OUT/general_phase_probes.txt || exit 1
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/tmp
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/tmp -- python3 tools/general_rate.py redsec_medium > $OUT/pmc_$C.log 2>&1 || exit 1
  f=$(find $OUT/tmp -name "*counter_collection.csv" | head -1)
  grep -E "Counter_Name|gen_blind_rotate" "$f" > $OUT/medium_$C.csv
done
rm -rf $OUT/tmp
head -5 $OUT/medium_FETCH_SIZE.csv
