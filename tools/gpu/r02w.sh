cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in pnomerge; do
  echo "== $v"
  timeout 300 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_$v.so 2>&1 | grep "step"
  timeout 300 python tools/sweep.py --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_$v.so 2>&1 | grep "step"
  timeout 300 python tools/sweep.py --op phrase --terms 2 --k 100 --docs 6250000 --scorer tfidf --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_$v.so 2>&1 | grep "step"
done > gpurun_out/r02w.txt 2>&1
echo "== base 6.25M tfidf" >> gpurun_out/r02w.txt
timeout 300 python tools/sweep.py --op phrase --terms 2 --k 100 --docs 6250000 --scorer tfidf --configs 8192:64 --nocheck --touched 2>&1 | grep "step\|touched" >> gpurun_out/r02w.txt
cat gpurun_out/r02w.txt
