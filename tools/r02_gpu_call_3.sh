cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python bench.py --cpu-baseline off --steps 24 --warmup 8 > gpurun_out/r02/bench_gemm_dma.json 2> gpurun_out/r02/bench_gemm_dma.err
UPAMD_GEMM_NT_DMA=0 timeout 600 python bench.py --cpu-baseline off --steps 24 --warmup 8 > gpurun_out/r02/bench_gemm_v0.json 2> gpurun_out/r02/bench_gemm_v0.err
python - <<'PY'
import json
for f in ('bench_gemm_dma','bench_gemm_v0'):
    d=json.loads(open('gpurun_out/r02/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['value']), d['ms_per_step'], d['roofline']['achieved'], d['kernel_ms_per_step'])
PY
