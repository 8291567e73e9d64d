cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04t; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_tiny.py -m gpu -q -x 2>&1 | tail -3) > $O/gpu_tests_tiny.log 2>&1
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 256 --warmup 256 > $O/bench_hlg_ref.json 2>/dev/null
timeout 300 python bench.py --workload grid_ref --cpu-baseline off --steps 100 --warmup 200 > $O/bench_grid_ref.json 2>/dev/null
timeout 300 python tools/lab_trace/trace_tiny.py hlg_ref > $O/trace_hlg_ref.log 2>&1
tail -2 $O/gpu_tests_tiny.log
grep -A10 "phase timeline" $O/trace_hlg_ref.log | cut -c1-100
for f in $O/bench_hlg_ref.json $O/bench_grid_ref.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d.get('kernel_ms_per_step'))
PY
done
