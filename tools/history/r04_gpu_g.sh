# round 4: fused small-model kernel with the LDS left over given to the head's chunk buffers; finer section profile
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04g; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_tiny.py tests/test_gpu_deep_edge.py -m gpu -q -x 2>&1 | tail -5) > $O/gpu_tests_tiny.log 2>&1
OMP_NUM_THREADS=1 timeout 300 python tools/r04_diag_tiny.py > $O/diag.log 2>&1
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 256 --warmup 256 > $O/bench_hlg_ref.json 2>/dev/null
timeout 300 python bench.py --workload grid_ref --cpu-baseline off --steps 100 --warmup 200 > $O/bench_grid_ref.json 2>/dev/null
tail -3 $O/gpu_tests_tiny.log; grep -A40 "fused kernel sections" $O/diag.log
for f in $O/bench_*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d.get('kernel_ms_per_step'))
PY
done
