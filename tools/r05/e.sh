# round 5, call e: rollout serving after the lean path (+ a cProfile of in-process 64-row rounds), bench lines with the packer
# changes (default thread count, records leg), breakdown at the default thread count.  Tight timeouts everywhere.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05e; mkdir -p $O
(timeout 240 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x 2>&1 | tail -12) > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log
if grep -q failed $O/gpu_tests.log; then echo "rollout tests failed: stopping"; exit 1; fi
timeout 200 python tools/rollout_bench.py --D 16 --L 2 --profile 50 > $O/rollout_d16.json 2> $O/rollout_d16.err
timeout 200 python tools/rollout_bench.py --D 256 --L 3 --clients 16 64 --cpu-procs 1 16 --cpu-requests 10 > $O/rollout_d256.json 2> $O/rollout_d256.err
timeout 120 python tools/inclusive_breakdown.py --workload hlg_ref --unique > $O/breakdown_hlg_ref.json 2>> $O/breakdown.err
timeout 160 python tools/inclusive_breakdown.py --workload hlg_d256 --unique > $O/breakdown_hlg_d256.json 2>> $O/breakdown.err
timeout 160 python bench.py --workload hlg_ref --steps 256 --warmup 256 --cpu-baseline off > $O/bench_hlg_ref.json 2> $O/bench_hlg_ref.err
timeout 240 python bench.py --steps 20 --warmup 5 --cpu-baseline off > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/breakdown_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], {k: round(v,1) for k,v in d.items() if k.startswith('call_')}, {k: round(v,1) for k,v in d['phases_ms'].items()})
    except Exception as e: print(f, 'FAILED', e)
for f in sorted(glob.glob('$O/rollout_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['model'], d.get('inprocess_ms_per_64_row_batch'), d['cpu_select_action'])
        for s in d['serving']: print('   ', {k: (round(v,2) if isinstance(v,float) else v) for k,v in s.items()})
    except Exception as e: print(f, 'FAILED', e)
for f in sorted(glob.glob('$O/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); u=d['update_params_inclusive']; r=d.get('update_params_inclusive_records') or {}
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), (d.get('strong_proxy') or {}).get('value'), 'incl', round(u['samples_per_s']), round(u['fraction_of_step_rate'],3), round(u['prepare_s'],4), 'records', round(r.get('samples_per_s',0)), round(r.get('fraction_of_step_rate',0),3), round(r.get('prepare_s',0),4))
    except Exception as e: print(f, 'FAILED', e)
PY
grep -A34 "cumulative" $O/rollout_d16.err | head -44; tail -n 2 $O/*.err | tail -24
