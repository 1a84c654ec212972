cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "join_edge or paths_agree or pilot_misled or queries_ragged or config3_or8" > $O/${TAG:-r03e}_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/${TAG:-r03e}_tests.log
timeout 800 python tools/join_tune.py --runs ${RUNS:-base:1024} > $O/${TAG:-r03e}_tune.log 2>&1; echo "tune rc=$?"; grep -v amdgpu.ids $O/${TAG:-r03e}_tune.log | tail -8
