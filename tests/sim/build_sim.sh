#!/bin/sh
# tests/sim/build_sim.sh — TEST INFRASTRUCTURE ONLY: compiles the product sources
# (unmodified) against the CPU fiber emulator into tests/sim/libirs_hip_sim.so.
set -e
cd "$(dirname "$0")/../.."
exec g++ -O1 -g -std=c++17 -fPIC -shared -pthread -ffp-contract=off \
  -I include -I iresearch_amd/csrc -I tests/sim -I iresearch_amd/csrc/hip \
  -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-maybe-uninitialized \
  -o tests/sim/libirs_hip_sim.so -x c++ iresearch_amd/csrc/irs_hip.hip \
  -x assembler tests/sim/sim_switch.S
