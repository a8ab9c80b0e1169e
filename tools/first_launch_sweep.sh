#!/bin/bash
# N fresh processes, one after the other, each tests/first_launch_stress.py with its own seed (the -m gpu suite runs six):
# first launches of every kernel form + a full-size redsec_params_medium batch run twice. Prints one line per process and a total.
# usage: tools/first_launch_sweep.sh [N=24] [seed_base=500]
N="${1:-24}"; S="${2:-500}"; bad=0
for i in $(seq 1 "$N"); do
  out=$(timeout -k 10 240 python tests/first_launch_stress.py $((S + i)) 2>/dev/null | grep '^{' | tail -1)
  echo "$out" | cut -c1-200
  echo "$out" | grep -q '"findings": \[\]' || bad=$((bad + 1))
done
echo "fresh processes with a finding: $bad of $N"
[ "$bad" -eq 0 ]
