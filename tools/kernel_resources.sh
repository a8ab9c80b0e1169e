#!/bin/bash
# Compact per-kernel resource table (VGPRs, AGPRs, scratch, occupancy, LDS) for a .hip file.
# usage: tools/kernel_resources.sh redsec_amd/csrc/rs_bootstrap.hip [filter]
cd "$(dirname "$0")/.."
EXTRA=""; case "$1" in *rs_bootstrap.hip) EXTRA="-mllvm -enable-post-misched=0";; esac   # as redsec_amd/build.py HIP_OBJECTS
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude -Iredsec_amd/csrc $EXTRA --cuda-device-only -c "$1" -o /tmp/kr.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|Name:|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/^[^ ]+ remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | grep -E "${2:-.}" | c++filt | sed -E 's/rs::BlindRotateArgs//; s/rs:://g'
