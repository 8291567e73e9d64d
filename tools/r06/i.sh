# round 6, call i: evidence pass 1 again on the FINAL csrc (the packer plans SGNN replays without the rl-mlp sections), the
# reference-dims lines and their update_params breakdown, the prepare() chunk experiment at the YAML dims, serving, then the flake loop
ROUND=r06 bash tools/evidence/main.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06
timeout 300 python bench.py --workload hlg_ref --steps 256 --warmup 256 > $O/bench_hlg_ref.json 2>/dev/null
timeout 300 python bench.py --workload grid_ref --steps 100 --warmup 200 > $O/bench_grid_ref.json 2>/dev/null
for c in 1 2 4; do UPAMD_PREPARE_CHUNKS=$c timeout 300 python bench.py --workload hlg_ref --steps 64 --warmup 256 --cpu-baseline off > $O/bench_hlg_ref_chunks$c.json 2>/dev/null; done
timeout 300 python tools/inclusive_breakdown.py --workload hlg_ref --unique > $O/breakdown_hlg_ref.json 2> $O/breakdown_hlg_ref.err
timeout 300 python tools/inclusive_breakdown.py --workload hlg_d256 --unique > $O/breakdown_hlg_d256.json 2> $O/breakdown_hlg_d256.err
timeout 300 python tools/rollout_bench.py --D 16 --L 2 > $O/rollout_d16.json 2> $O/rollout_d16.err
timeout 300 python tools/rollout_bench.py --D 256 --L 3 --clients 8 16 32 64 --cpu-procs 1 16 --cpu-requests 10 > $O/rollout_d256.json 2> $O/rollout_d256.err
python tools/evidence/lines.py $O/bench_hlg_ref*.json $O/bench_grid_ref.json
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/rollout_*.json'))+sorted(glob.glob('$O/breakdown_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if 'serving' in d:
            for s in d['serving']: print(f.split('/')[-1], {k: (round(v,2) if isinstance(v,float) else v) for k,v in s.items() if k!='window'})
        else: print(f.split('/')[-1], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,dict)}, d.get('phases_ms'))
    except Exception as e: print(f, 'FAILED', e)
PY
AMD_LOG_LEVEL=1 bash tools/r06/flake_loop.sh 1 ${FLAKE_BUDGET:-1500}
cat gpurun_out/flake/serving_stats.jsonl 2>/dev/null | tail -12
