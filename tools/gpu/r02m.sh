cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 --touched 2>&1 | grep "step\|Error\|error\|touched\|Assert"
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --touched 2>&1 | grep "step\|Error\|error\|touched"
  timeout 300 python tools/sweep.py --op and --terms 4 --configs 8192:64 --nocheck 2>&1 | grep "step\|Error\|error" ) > gpurun_out/r02m.txt 2>&1
cat gpurun_out/r02m.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02m_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 gpurun_out/r02m_gputests.log
