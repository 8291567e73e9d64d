# round 5, call b: new GPU tests (bound agent rollout / checkpoints, pipelined prepare), the inclusive-call breakdown at both
# model sizes, default + reference-dims bench lines with the pipelined prepare
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05b; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_update_branches.py tests/test_gpu_parity.py tests/test_gpu_dp.py -m gpu -q -x --durations=5 2>&1 | tail -25) > $O/gpu_tests.log 2>&1
for w in hlg_ref hlg_d256; do
  timeout 600 python tools/inclusive_breakdown.py --workload $w --unique > $O/breakdown_${w}_unique.json 2> $O/breakdown_${w}.err
  timeout 600 python tools/inclusive_breakdown.py --workload $w > $O/breakdown_${w}_pool.json 2>> $O/breakdown_${w}.err
done
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline off > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --workload hlg_ref --steps 256 --warmup 256 --cpu-baseline off > $O/bench_hlg_ref.json 2> $O/bench_hlg_ref.err
UPAMD_PREPARE_CHUNKS=1 timeout 600 python bench.py --workload hlg_ref --steps 256 --warmup 256 --cpu-baseline off > $O/bench_hlg_ref_chunks1.json 2>/dev/null
timeout 600 python bench.py --workload grid_ref --steps 100 --warmup 200 --cpu-baseline off > $O/bench_grid_ref.json 2>/dev/null
tail -8 $O/gpu_tests.log; cat $O/breakdown_*.json; tail -3 $O/breakdown_*.err
for f in $O/bench*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); u=d['update_params_inclusive']
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), (d.get('strong_proxy') or {}).get('value'), (d.get('strong_proxy') or {}).get('ms_share'), round(u['samples_per_s']), round(u['fraction_of_step_rate'],3), round(u['prepare_s'],4), round(u['loop_s'],4), u['unique_host_states'])
except Exception as e: print('$f', 'FAILED', e)
PY
done
