#!/bin/sh
# tools/build_variant.sh NAME [SED_EXPR...] — builds (from $IRS_SRC, default the tree's csrc) gpurun_variants/libirs_hip_NAME.so from a
# scratch copy of iresearch_amd/csrc with the given sed expressions applied to every source
# (tuning experiments: constants changed, parts compiled out).  gpurun_variants/ is git-ignored
# but travels to the GPU box; tools/gpu/*.sh copy a variant over csrc/libirs_hip.so.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
TMP="$(mktemp -d /tmp/irsvar.XXXXXX)"
cp -r "${IRS_SRC:-$ROOT/iresearch_amd/csrc}" "$TMP/csrc"
rm -f "$TMP"/csrc/*.so
for e in "$@"; do sed -i -e "$e" "$TMP"/csrc/*.h "$TMP"/csrc/*.hip; done
mkdir -p "$ROOT/gpurun_variants"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc \
  -Wno-unused-function -I "$ROOT/include" -I "$TMP/csrc" -I "$TMP/csrc/hip" \
  -o "$ROOT/gpurun_variants/libirs_hip_$NAME.so" "$TMP"/csrc/*.hip
rm -rf "$TMP"
echo "built gpurun_variants/libirs_hip_$NAME.so"
