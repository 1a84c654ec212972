# tools/gpu/conj_abl.sh — conjunction sweeps under ablation builds (gpurun_variants/libirs_hip_NAME.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/conj_abl.log; : > $O
for T in ${TERMS:-2 3}; do
  for V in ${VARIANTS:-base c1 c2 c3}; do
    LIB=""; [ $V != base ] && LIB="--lib gpurun_variants/libirs_hip_$V.so"
    echo "== and $T $V" >> $O
    timeout 300 python tools/sweep.py --op and --terms $T --configs 8192:64 --path items --nocheck $LIB 2>&1 | grep -E "tile=" >> $O
  done
done
echo "== or 3 clustered wand" >> $O
timeout 300 python tools/sweep.py --op or --terms 3 --clustered --wand --configs 8192:64 2>&1 | grep -E "tile=|path|WAND|hits" >> $O
cat $O
