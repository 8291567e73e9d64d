cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python tools/gemm_lab.py > gpurun_out/r02/gemm_lab_7.log 2>&1
cat gpurun_out/r02/gemm_lab_7.log
