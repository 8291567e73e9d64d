# round 6, call c: the rollout side after the ring transport + upamd_select_actions: its tests, serving throughput / latency
# (tools/rollout_bench.py, both model sizes; the torch route of round 5 as the A/B), a cProfile of the in-process 64-row round
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06c; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -40) > $O/rollout_tests.log 2>&1; tail -3 $O/rollout_tests.log
timeout 300 python tools/rollout_bench.py --D 16 --L 2 --profile 100 > $O/rollout_d16.json 2> $O/rollout_d16.err
timeout 300 python tools/rollout_bench.py --D 256 --L 3 --clients 8 16 32 64 --cpu-procs 1 16 --cpu-requests 10 --profile 100 > $O/rollout_d256.json 2> $O/rollout_d256.err
UPAMD_SERVE_SELECT=torch timeout 300 python tools/rollout_bench.py --D 16 --L 2 --cpu-procs 1 > $O/rollout_d16_torch_select.json 2> $O/rollout_d16_torch_select.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/rollout_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['model'], d.get('inprocess_ms_per_64_row_batch'), d['cpu_select_action'])
        for s in d['serving']: print('   ', {k: (round(v,2) if isinstance(v,float) else v) for k,v in s.items()})
    except Exception as e: print(f, 'FAILED', e)
PY
grep -A32 "cumulative" $O/rollout_d16.err | head -45
