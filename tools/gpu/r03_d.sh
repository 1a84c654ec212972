cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 800 python tools/join_tune.py --runs ${RUNS:-base:1024} > $O/${TAG:-r03d}_tune.log 2>&1; echo "tune rc=$?"; grep -v amdgpu.ids $O/${TAG:-r03d}_tune.log | tail -12
