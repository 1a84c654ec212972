cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 2>&1 | grep "step\|Error\|error\|touched\|Assert"
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 2>&1 | grep "step\|Error\|error\|touched"
  timeout 300 python tools/sweep.py --op and --terms 4 --configs 8192:64 --nocheck 2>&1 | grep "step\|Error\|error" ) > gpurun_out/r02n.txt 2>&1
cat gpurun_out/r02n.txt
