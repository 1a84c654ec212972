# round 3, first GPU session: the joined-stream path on real hardware (parity first, then timings)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "paths_agree or pilot_misled or config3_or8 or config2 or queries_all_scorers or queries_ragged or no_norms or multi_segment_batch" > $O/r03a_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r03a_tests.log
timeout 900 python tools/join_tune.py --runs base:items,base:512,base:1024,w0p8:512,w8p4:1024,w8p4:512,base:256 > $O/r03a_tune.log 2>&1; echo "tune rc=$?"; cat $O/r03a_tune.log | tail -12
