#!/bin/bash
# Build libredsec_hip.so from a source tree into variants/lib_<name>.so (git-ignored, travels to the GPU
# box) for same-box A/B timing with REDSEC_HIP_LIB. usage: tools/build_variant.sh NAME [SRC_ROOT] [extra hipcc flags...]
# Same recipe as redsec_amd/build.py: one object per source, rs_bootstrap.hip with $BS_FLAGS on top (default: the
# post-register-allocation scheduler off, as build.py's HIP_FILE_FLAGS; BS_FLAGS=" " builds it with the common flags only).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; SRC="${2:-$ROOT}"; shift; shift || true
BS_FLAGS="${BS_FLAGS:--mllvm -enable-post-misched=0}"   # rs_bootstrap.hip only
GEN_FLAGS="${GEN_FLAGS:-}"                               # rs_general.hip only (A/B)
mkdir -p "$ROOT/variants"
OBJ="$(mktemp -d /tmp/variant_$NAME.XXXX)"
COMMON="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -I$SRC/include -I$SRC/redsec_amd/csrc"
pids=()
for f in rs_bootstrap.hip rs_general.hip rs_kernels.hip rs_api.cpp; do
  extra=""; [ "$f" = rs_bootstrap.hip ] && extra="$BS_FLAGS"; [ "$f" = rs_general.hip ] && extra="$GEN_FLAGS"
  hipcc $COMMON $extra "$@" -c "$SRC/redsec_amd/csrc/$f" -o "$OBJ/${f%.*}.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -o "$ROOT/variants/lib_$NAME.so"
rm -rf "$OBJ"
ls -la "$ROOT/variants/lib_$NAME.so"
