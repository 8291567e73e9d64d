cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -60) > gpurun_out/r02/gpu_tests_3.log 2>&1; bash tools/r02_gpu_call_4.sh > gpurun_out/r02/call4.log 2>&1; tail -75 gpurun_out/r02/call4.log | cut -c1-170
tail -40 gpurun_out/r02/gpu_tests_3.log
