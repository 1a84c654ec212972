cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python tools/join_tune.py --runs base:items,base:1024,base:512 > $O/r03c_tune.log 2>&1; echo "tune rc=$?"; tail -4 $O/r03c_tune.log
