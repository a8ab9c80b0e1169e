#!/bin/bash
# first-launch scenario: fresh processes, each one medium batch on a synthetic key; a wrong product trips the enforced certificate
OUT=gpurun_out/r03_race; mkdir -p $OUT
for v in gen_racy gen_fixed; do
  bad=0
  for i in $(seq 1 24); do
    REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 120 python tools/r03/r03_race_repro.py redsec_medium 1024 2 2>&1 | grep -q "certificate: 0" || bad=$((bad+1))
  done
  echo "$v: fresh processes with a refused or differing run: $bad of 24" | tee -a $OUT/race_first_launch.txt
done
