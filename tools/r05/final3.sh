# round 5, last pass on the final tree: the whole GPU suite (incl. the RCCL single-rank bucket test), the default line once more
# (another box), the default line under a one-rank RCCL group with the buckets forced on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05x; mkdir -p $O
(timeout 1000 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 420 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
UPAMD_DIST_FORCE_INIT=1 UPAMD_GRAD_BUCKETS=force RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off --inclusive-pool --strong-proxy off > $O/bench_rccl_single_rank_buckets_forced.json 2> $O/rccl_forced.log
UPAMD_DIST_FORCE_INIT=1 UPAMD_GRAD_BUCKETS=force RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout 300 python bench.py --minibatch 256 --steps 96 --warmup 32 --cpu-baseline off --inclusive-pool --no-kernel-events > $O/bench_rccl_single_rank_buckets_forced_mb256.json 2>> $O/rccl_forced.log
UPAMD_DIST_FORCE_INIT=1 UPAMD_GRAD_BUCKETS=0 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 timeout 300 python bench.py --minibatch 256 --steps 96 --warmup 32 --cpu-baseline off --inclusive-pool --no-kernel-events > $O/bench_rccl_single_rank_single_mb256.json 2>> $O/rccl_forced.log
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; u=d['update_params_inclusive']
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round(r.get('frac') or 0,4), r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'), (d.get('strong_proxy') or {}).get('value'), round(u['fraction_of_step_rate'],3), (d.get('message_passing') or {}).get('valu_busy'))
except Exception as e: print('$f', 'FAILED', e)
PY
done; tail -3 $O/rccl_forced.log
