cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r03w
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5) > gpurun_out/r03w/gpu_tests.log 2>&1
tail -3 gpurun_out/r03w/gpu_tests.log
