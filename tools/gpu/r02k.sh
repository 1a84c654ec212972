cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in cbase cleadonly cnodecode cnoput; do
  echo "== $v"
  timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_$v.so 2>&1 | grep "step"
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_$v.so 2>&1 | grep "step"
done > gpurun_out/r02k.txt 2>&1
cat gpurun_out/r02k.txt
