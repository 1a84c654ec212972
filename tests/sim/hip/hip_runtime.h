// tests/sim/hip/hip_runtime.h — TEST INFRASTRUCTURE ONLY.
// Stands in for <hip/hip_runtime.h> when the product sources are compiled by
// g++ against the CPU fiber emulator (hip_sim.h) in the CPU-only test tier.
#pragma once
#include "../hip_sim.h"
