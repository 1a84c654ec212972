#!/bin/sh
# tools/isa_one.sh SRC.hip OUT.s — device assembly of one translation unit (gfx950), with the
# product's compile flags; feed OUT.s to tools/isa_stats.py.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -Wno-unused-function \
  -I "$ROOT/include" -I "$ROOT/iresearch_amd/csrc" -I "$ROOT/iresearch_amd/csrc/hip" \
  -S --cuda-device-only -o "$2" "$1"
