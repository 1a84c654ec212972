cd $GRAFT_REPO_ROOT
timeout 300 python tools/fuzz_parity.py --seconds 120 --seed 41 2>&1 | tail -3
timeout 300 python tools/fuzz_parity.py --seconds 60 --seed 42 --max-docs 2000000 2>&1 | tail -3
