# lab: LDS-DMA gemm_nt with a bigger block per wave (2 waves per SIMD) and s_setprio around the MFMA burst
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
GEMM_LAB_VARIANTS=1,5,6,7,8,9,1 timeout 600 python tools/gemm_lab.py > $O/gemm_lab_wave_tiles.log 2>&1
cat $O/gemm_lab_wave_tiles.log
