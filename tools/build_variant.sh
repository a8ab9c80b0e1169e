#!/bin/bash
# Build libredsec_hip.so from a source tree into variants/lib_<name>.so (git-ignored, travels to the GPU
# box) for same-box A/B timing with REDSEC_HIP_LIB. usage: tools/build_variant.sh NAME [SRC_ROOT] [extra hipcc flags...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; SRC="${2:-$ROOT}"; shift; shift || true
mkdir -p "$ROOT/variants"
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -I"$SRC/include" -I"$SRC/redsec_amd/csrc" "$@" \
  "$SRC/redsec_amd/csrc/rs_bootstrap.hip" "$SRC/redsec_amd/csrc/rs_general.hip" "$SRC/redsec_amd/csrc/rs_kernels.hip" "$SRC/redsec_amd/csrc/rs_api.cpp" \
  -o "$ROOT/variants/lib_$NAME.so" 2>&1 | grep -v "hip-link" || true
ls -la "$ROOT/variants/lib_$NAME.so"
