#!/bin/bash
# Same-box A/B of library variants on the latency forms: tools/ab_small_batches.sh OUT ROUNDS name1 name2 ...
# (variants/lib_<name>.so, tools/build_variant.sh); per variant and round: tools/small_batch_probe.py on both shipped sets.
set -o pipefail
cd "$(dirname "$0")/.."
OUT="$1"; ROUNDS="$2"; shift; shift
for r in $(seq "$ROUNDS"); do for v in "$@"; do
  echo "== $v round $r"
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 120 python tools/small_batch_probe.py 196 256 --reps 9 2>/dev/null || exit 1
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 120 python tools/small_batch_probe.py 196 --reps 9 --params default128 2>/dev/null || exit 1
done; done > "$OUT" 2>&1
grep "==\|B 196" "$OUT"
