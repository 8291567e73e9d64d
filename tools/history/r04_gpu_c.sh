cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_tiny.py -m gpu -x -q 2>&1 | tail -15) > $O/tests_tiny.log 2>&1
tail -3 $O/tests_tiny.log
for t in 1024 512; do
UPAMD_TUNE=tiny_threads=$t timeout 300 python tools/r04_diag_tiny.py hlg_ref > $O/diag_$t.log 2>&1
grep -A25 "fused kernel sections" $O/diag_$t.log; grep "epoch" $O/diag_$t.log
done
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 32 --warmup 8 > $O/bench_hlg_ref.json 2>/dev/null
timeout 300 python bench.py --workload grid_ref --cpu-baseline off --steps 10 --warmup 4 > $O/bench_grid_ref.json 2>/dev/null
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d.get('kernel_ms_per_step'))
except Exception as e:
    print('$f FAILED', e)
PY
done
