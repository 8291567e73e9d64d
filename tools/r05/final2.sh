# round 5, final evidence (2/2): the other BASELINE workloads, the 256-row step, RCCL single rank, 2-rank functional lines
# (bucketed / single collective), rollout serving
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05y; mkdir -p $O
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x > $O/rollout_tests_$i.log 2>&1; tail -2 $O/rollout_tests_$i.log; done
grep -h -B30 "^E " $O/rollout_tests_*.log | head -80
(timeout 300 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_tiny.py tests/test_gpu_mlp.py -m gpu -q 2>&1 | tail -30) > $O/rollout_after_tiny.log 2>&1; tail -3 $O/rollout_after_tiny.log
for w in hlg_concept_d256 dhm_d256 mixed_d256; do
  timeout 300 python bench.py --workload $w --cpu-baseline off --steps 8 --warmup 3 --inclusive-pool > $O/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --workload hlg_ref --steps 256 --warmup 256 > $O/bench_hlg_ref.json 2>/dev/null
timeout 300 python bench.py --workload grid_ref --steps 100 --warmup 200 > $O/bench_grid_ref.json 2>/dev/null
timeout 200 python bench.py --minibatch 256 --cpu-baseline off --steps 64 --warmup 16 --inclusive-pool > $O/bench_hlg_d256_minibatch256.json 2>/dev/null
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off --inclusive-pool > $O/bench_rccl_single_rank.json 2> $O/rccl_single_rank.log
R="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 12 --warmup 4 --minibatch 512 --inclusive-pool"
UPAMD_DIST_BACKEND=gloo UPAMD_GRAD_BUCKETS=force timeout 300 $R > $O/bench_2ranks_1gpu_gloo_bucketed.json 2> $O/r2b.err
UPAMD_DIST_BACKEND=gloo UPAMD_GRAD_BUCKETS=0 timeout 300 $R > $O/bench_2ranks_1gpu_gloo_single.json 2> $O/r2s.err
timeout 200 python tools/rollout_bench.py --D 16 --L 2 > $O/rollout_d16.json 2> $O/rollout_d16.err
timeout 200 python tools/rollout_bench.py --D 256 --L 3 --clients 8 16 32 64 --cpu-procs 1 16 --cpu-requests 10 > $O/rollout_d256.json 2> $O/rollout_d256.err
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; u=d['update_params_inclusive']
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round(r.get('frac') or 0,4), (d.get('cpu_baseline') or {}).get('value'), d.get('allreduce_ms'), round(u['fraction_of_step_rate'],3), u['unique_host_states'], (d.get('update_params_inclusive_records') or {}).get('fraction_of_step_rate'))
except Exception as e: print('$f', 'FAILED', e)
PY
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/rollout_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['model'], d['cpu_select_action'])
        for s in d['serving']: print('   ', {k: (round(v,2) if isinstance(v,float) else v) for k,v in s.items()})
    except Exception as e: print(f, 'FAILED', e)
PY
