cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -60) > gpurun_out/r02/gpu_tests_3.log 2>&1
tail -40 gpurun_out/r02/gpu_tests_3.log
