# round 5, last call: the whole GPU suite on the final tree, exactly as the driver runs it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05m; mkdir -p $O
(timeout 1000 python -m pytest tests/ -x -q -m gpu --durations=6 2>&1 | tail -14) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
