# tools/gpu/tune.sh — A/B timings of the headline batch under several builds / settings, one
# index build (tools/join_tune.py):  RUNS=base:items,base:1024,NAME:512 TAG=r03x bash tools/gpu/tune.sh
# (NAME = gpurun_variants/libirs_hip_NAME.so from tools/build_variant.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 800 python tools/join_tune.py --runs ${RUNS:-base:items,base:1024} > $O/${TAG:-tune}_tune.log 2>&1; echo "tune rc=$?"; grep -v amdgpu.ids $O/${TAG:-tune}_tune.log | tail -12
