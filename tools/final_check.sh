#!/bin/bash
# The last GPU step of a round: the whole `-m gpu` suite on a fresh MI355X box, its log committed under profiles/rNN/ with a
# header naming the commit and the code-tree hash it ran on. tests/test_final_check_cpu.py fails in `-m "not gpu"` when the
# tree's code differs from the newest such log: documentation, profiles and tools may change after this run, code may not.
#
# usage (in the development container, not on the GPU box): tools/final_check.sh [rNN] [--with-bench]
#   refuses to run on a tree whose code paths have uncommitted changes (the log must name a commit that holds what ran)
set -o pipefail
cd "$(dirname "$0")/.."
ROUND="${1:-r06}"; [[ "$ROUND" == --* ]] && ROUND=r06
WITH_BENCH=0; for a in "$@"; do [ "$a" == "--with-bench" ] && WITH_BENCH=1; done
GPURUN=/usr/local/graft/bin/gpurun

dirty=$(git status --short -- redsec_amd include oracle tests bench.py __graft_entry__.py)
if [ -n "$dirty" ]; then echo "uncommitted code changes -- commit first:"; echo "$dirty"; exit 2; fi
head=$(git rev-parse HEAD)
tree=$(python tools/tree_hash.py) || exit 2
python -m redsec_amd.build > /dev/null || { echo "build failed"; exit 2; }     # the .so files travel with the snapshot

label="final_${head:0:10}"
cmd="timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/${label}_suite.log 2>&1; rc=\$?; tail -5 gpurun_out/${label}_suite.log; python tools/tree_hash.py > gpurun_out/${label}_tree_on_box.txt; [ \$rc -eq 0 ]"
if [ $WITH_BENCH -eq 1 ]; then
  cmd="$cmd && timeout -k 10 600 python bench.py > gpurun_out/${label}_bench.log 2> gpurun_out/${label}_bench.err && tail -1 gpurun_out/${label}_bench.log > gpurun_out/${label}_bench.json"
fi
mkdir -p gpurun_out "profiles/$ROUND"
rm -f "gpurun_out/${label}_suite.log" "gpurun_out/${label}_tree_on_box.txt"
$GPURUN --timeout 1200 -- "$cmd"
grc=$?
[ -f "gpurun_out/${label}_suite.log" ] || { echo "no suite log came back (gpurun rc $grc)"; exit 3; }
box_tree=$(cat "gpurun_out/${label}_tree_on_box.txt" 2>/dev/null)
log="profiles/$ROUND/gpu_suite_${label}.log"
{
  echo "# final_check: python -m pytest tests -m gpu -x -q on a fresh MI355X box (gpurun rc $grc)"
  echo "# date: $(date -u +%Y-%m-%dT%H:%M:%SZ)"
  echo "# head: $head"
  echo "# git status --short (code paths): clean"
  echo "# tree_sha256: $tree"
  echo "# tree_sha256_on_box: $box_tree"
  echo "# result: $(tail -1 "gpurun_out/${label}_suite.log")"
  cat "gpurun_out/${label}_suite.log"
} > "$log"
echo "wrote $log"
if [ $WITH_BENCH -eq 1 ] && [ -s "gpurun_out/${label}_bench.json" ]; then cp "gpurun_out/${label}_bench.json" "profiles/$ROUND/bench_line_${label}.json"; fi
[ "$box_tree" == "$tree" ] || { echo "tree hash on the box differs from the local one"; exit 4; }
exit $grc
