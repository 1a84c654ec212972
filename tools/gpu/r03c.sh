cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python tools/fuzz_parity.py --seconds 150 --seed 5 2>&1 | tail -5
timeout 400 python tools/fuzz_parity.py --seconds 120 --seed 6 --max-docs 3000000 2>&1 | tail -5
