#!/bin/bash
# Build libredsec_hip.so from a source tree into variants/lib_<name>.so (git-ignored, travels to the GPU
# box) for same-box A/B timing with REDSEC_HIP_LIB. usage: tools/build_variant.sh NAME [SRC_ROOT] [extra hipcc flags...]
# Same recipe as redsec_amd/build.py (HIP_OBJECTS): one object per source, rs_bootstrap.hip in its two parts, part 1 with
# $BS_FLAGS on top (default: the post-register-allocation scheduler off; BS_FLAGS=" " builds it with the common flags only).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; SRC="${2:-$ROOT}"; shift; shift || true
BS_FLAGS="${BS_FLAGS:--mllvm -enable-post-misched=0}"   # rs_bootstrap.hip part 1 only (FFT / NTT kernels, split duo form)
BS2_FLAGS="${BS2_FLAGS:--mllvm -amdgpu-sched-strategy=max-memory-clause}"                               # rs_bootstrap.hip part 2 only (split cooperative / lock-step kernels)
BS4_FLAGS="${BS4_FLAGS:-$BS_FLAGS}"                      # rs_bootstrap.hip part 4 only (coop8 with the listed step): part 1's flags by default
K_FLAGS="${K_FLAGS:-}"                                   # rs_kernels.hip only (A/B)
GEN_FLAGS="${GEN_FLAGS:-}"                               # rs_general.hip only (A/B)
mkdir -p "$ROOT/variants"
OBJ="$(mktemp -d /tmp/variant_$NAME.XXXX)"
COMMON="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -I$SRC/include -I$SRC/redsec_amd/csrc"
pids=()
cc() { hipcc $COMMON "$@" & pids+=($!); }
B="$SRC/redsec_amd/csrc"
if grep -q RS_BS_PART "$B/rs_bootstrap.hip"; then
  cc -DRS_BS_PART=1 $BS_FLAGS "$@" -c "$B/rs_bootstrap.hip" -o "$OBJ/rs_bootstrap_1.o"
  cc -DRS_BS_PART=2 $BS2_FLAGS "$@" -c "$B/rs_bootstrap.hip" -o "$OBJ/rs_bootstrap_2.o"
  if grep -q "RS_BS_PART & 4" "$B/rs_bootstrap.hip"; then cc -DRS_BS_PART=4 $BS4_FLAGS "$@" -c "$B/rs_bootstrap.hip" -o "$OBJ/rs_bootstrap_4.o"; fi
else   # an older source tree: one object
  cc $BS_FLAGS "$@" -c "$B/rs_bootstrap.hip" -o "$OBJ/rs_bootstrap.o"
fi
cc $GEN_FLAGS "$@" -c "$B/rs_general.hip" -o "$OBJ/rs_general.o"
cc $K_FLAGS "$@" -c "$B/rs_kernels.hip" -o "$OBJ/rs_kernels.o"
cc "$@" -c "$B/rs_api.cpp" -o "$OBJ/rs_api.o"
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -o "$ROOT/variants/lib_$NAME.so"
rm -rf "$OBJ"
ls -la "$ROOT/variants/lib_$NAME.so"
