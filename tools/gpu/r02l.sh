cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 2>&1 | grep "step\|Error\|error"
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck 2>&1 | grep "step\|Error\|error"
  timeout 300 python tools/sweep.py --op and --terms 4 --configs 8192:64 --nocheck 2>&1 | grep "step\|Error\|error" ) > gpurun_out/r02l.txt 2>&1
cat gpurun_out/r02l.txt
