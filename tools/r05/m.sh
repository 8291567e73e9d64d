# round 5, last call: the whole GPU suite on the final tree, exactly as the driver runs it (full output kept: a failure must be readable)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05m; mkdir -p $O
timeout 1000 python -m pytest tests/ -x -q -m gpu --durations=6 --tb=long > $O/gpu_tests_full.log 2>&1
tail -14 $O/gpu_tests_full.log > $O/gpu_tests.log; tail -4 $O/gpu_tests.log
grep -n "^E \|Error" $O/gpu_tests_full.log | head -40
