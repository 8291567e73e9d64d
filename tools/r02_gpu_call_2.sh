set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
tools/mfma_peak.bin > gpurun_out/r02/mfma_peak.log 2>&1
timeout 900 python tools/gemm_lab.py > gpurun_out/r02/gemm_lab_1.log 2>&1
(time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -60) > gpurun_out/r02/gpu_tests_2.log 2>&1
cat gpurun_out/r02/mfma_peak.log; cat gpurun_out/r02/gemm_lab_1.log; tail -5 gpurun_out/r02/gpu_tests_2.log
