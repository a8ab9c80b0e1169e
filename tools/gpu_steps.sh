#!/bin/bash
# One gpurun call = a list of named steps, logs under gpurun_out/<label>_<step>.*; stops at the first failing step.
# usage (on the GPU box, through gpurun): tools/gpu_steps.sh LABEL step [step ...]
#   suite            python -m pytest tests -m gpu -x -q
#   tests:<expr>     python -m pytest <files / -k expression> -m gpu -x -q     (spaces as '+': tests:tests/test_gpu_cifar.py+-k+fused)
#   bench[:args]     python bench.py [args]            -> <label>_bench.json (last line)
#   mnist[:ENV=1]    tools/mnist_latency.py, optionally with one environment switch (same-box A/B: mnist mnist:RS_NO_COOP8=1)
#   prof[:args]      rocprofv3 --kernel-trace --stats of bench.py (bounded legs)  -> <label>_kernel_stats.csv
#   pmc:C1+C2/C3:script+args   rocprofv3 --pmc passes (one run per '/' group) of python <script>, folded per kernel by tools/pmc_fold.py
#   py:<script+args> python <script> <args>
set -o pipefail
cd "$(dirname "$0")/.."
LABEL="$1"; shift
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
for step in "$@"; do
  name="${step%%:*}"; arg=""; [[ "$step" == *:* ]] && arg="${step#*:}"; arg="${arg//+/ }"
  tag="${LABEL}_$(echo "$step" | tr -c 'A-Za-z0-9_.=\n' '_' | cut -c1-48)_$(echo "$step" | md5sum | cut -c1-5)"   # unique per step
  echo "=== $step ($(date +%T))"
  case "$name" in
    suite) timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/${LABEL}_suite.log 2>&1; rc=$?; tail -25 $OUT/${LABEL}_suite.log;;
    tests) timeout -k 10 900 python -m pytest $arg -m gpu -x -q --durations=10 > $OUT/$tag.log 2>&1; rc=$?; tail -15 $OUT/$tag.log;;
    bench) timeout -k 10 600 python bench.py $arg > $OUT/$tag.log 2> $OUT/$tag.err; rc=$?; tail -1 $OUT/$tag.log > $OUT/$tag.json; tail -c 3000 $OUT/$tag.json; echo; tail -3 $OUT/$tag.err;;
    mnist) if [ -n "$arg" ]; then env $arg timeout -k 10 300 python tools/mnist_latency.py > $OUT/$tag.txt 2>&1; else timeout -k 10 300 python tools/mnist_latency.py > $OUT/$tag.txt 2>&1; fi; rc=$?; cat $OUT/$tag.txt;;
    prof) rm -rf /tmp/prof_$LABEL; (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$LABEL -- python3 $OLDPWD/bench.py --cpu-sample 0 --no-live-traffic --no-cifar $arg > $OLDPWD/$OUT/$tag.log 2>&1); rc=$?
          f=$(find /tmp/prof_$LABEL -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/${LABEL}_kernel_stats.csv && head -12 $OUT/${LABEL}_kernel_stats.csv;;
    pmc) # pmc:COUNTERS[/COUNTERS...]:script+args -- one rocprofv3 --pmc run per '/'-separated counter group (counters only)
          groups="${arg%%:*}"; cmd="${arg#*:}"; rc=0
          IFS='/' read -ra GS <<< "$groups"
          for g in "${GS[@]}"; do
            rm -rf /tmp/pmc_$LABEL; gt=$(echo $g | tr ' ' '_' | cut -c1-40)
            timeout -k 10 600 rocprofv3 --pmc $g --output-format csv -d /tmp/pmc_$LABEL -- python3 $cmd > $OUT/${LABEL}_pmc_$gt.log 2>&1 || rc=$?
            f=$(find /tmp/pmc_$LABEL -name '*counter_collection.csv' | head -1)
            [ -n "$f" ] && python3 tools/pmc_fold.py "$f" > $OUT/${LABEL}_pmc_$gt.txt && cat $OUT/${LABEL}_pmc_$gt.txt
          done;;
    py) envs=(); words=($arg); while [[ "${words[0]}" == *=* ]]; do envs+=("${words[0]}"); words=("${words[@]:1}"); done   # leading VAR=VALUE words: environment
        env "${envs[@]}" timeout -k 10 900 python "${words[@]}" > $OUT/$tag.txt 2>&1; rc=$?; tail -40 $OUT/$tag.txt;;
    *) echo "unknown step $step"; rc=2;;
  esac
  echo "=== $step rc=$rc"
  [ $rc -ne 0 ] && exit $rc
done
exit 0
