# round 5, call p: the data-parallel GPU tests after the bucket default became backend-dependent
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05p; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_update_branches.py tests/test_gpu_dp.py -m gpu -q -x --durations=4 2>&1 | tail -10) > $O/tests.log 2>&1; tail -6 $O/tests.log
