cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(timeout 600 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_rollout.py -q -x 2>&1 | tail -40) > gpurun_out/r02/gpu_tests_4.log 2>&1
tail -40 gpurun_out/r02/gpu_tests_4.log
