#!/bin/bash
# Same-box A/B: alternate bench.py over the variant libraries (variants/lib_<name>.so), ROUNDS times.
# usage: tools/ab_bench.sh ROUNDS "bench args" name1 name2 ...   (prints name, bootstraps/s, blind-rotate ms)
ROUNDS="$1"; ARGS="$2"; shift; shift
for r in $(seq "$ROUNDS"); do
  for n in "$@"; do
    REDSEC_HIP_LIB="$PWD/variants/lib_$n.so" python bench.py --cpu-sample 0 --no-exact-check --no-live-traffic $ARGS 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$ARGS', round(d['value']), d['kernels_ms'], 'ok' if d['checks']['all_outputs_decrypt_to_nand'] else 'WRONG')"
  done
done
