cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_kernels.py tests/test_gpu_gemm.py tests/test_gpu_mlp.py -m gpu -q -x 2>&1 | tail -25) > $O/gpu_tests.log 2>&1
tail -8 $O/gpu_tests.log
timeout 600 python bench.py --cpu-baseline off > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off > $O/bench_hlg_ref.json 2> $O/bench_hlg_ref.err
timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/bench_mb256.json 2>/dev/null
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --workload hlg_ref --cpu-baseline off --steps 40 --warmup 8 > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_ref.txt
for f in default hlg_ref mb256; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), d.get('roofline',{}).get('achieved'), d.get('kernel_ms_per_step'))
except Exception as e:
    print('$f', 'FAILED', e); print(open('$O/bench_$f.err').read()[-1500:] if '$f' != 'mb256' else '')
PY
done
head -30 $O/kernel_trace_hlg_ref.txt | cut -c1-140
