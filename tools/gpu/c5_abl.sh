# tools/gpu/c5_abl.sh — config 5 under ablation / experiment builds (gpurun_variants/libirs_hip_NAME.so):
# VARIANTS="base a b" bash tools/gpu/c5_abl.sh; every variant replaces csrc/libirs_hip.so on the
# box's scratch copy for its run.  Prints ms/step and the stage times of each.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/${TAG:-c5_abl}.log; : > $O
cp iresearch_amd/csrc/libirs_hip.so /tmp/libirs_hip_base.so
for V in ${VARIANTS:-base}; do
  if [ $V = base ]; then cp /tmp/libirs_hip_base.so iresearch_amd/csrc/libirs_hip.so
  else cp gpurun_variants/libirs_hip_$V.so iresearch_amd/csrc/libirs_hip.so; fi
  touch iresearch_amd/csrc/libirs_hip.so
  echo "== $V" >> $O
  timeout 600 python bench.py --config 5 --steps ${STEPS:-5} --warmup 3 --no-cpu 2>gpurun_out/${TAG:-c5_abl}_$V.err | \
    python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['roofline'].get('kernel_ms'), 'reruns', d['config'].get('reruns_rank0'))
" >> $O
done
cp /tmp/libirs_hip_base.so iresearch_amd/csrc/libirs_hip.so
cat $O
