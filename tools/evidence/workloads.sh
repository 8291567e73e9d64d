# Evidence pass 2 (GPU box): the other BASELINE workloads, the reference-dims workloads, the 256-row step (one rank's share of an
# 8-GPU strong-scaling step), a one-rank RCCL group, bench.py --gpus 2 started WITHOUT a launcher (gloo on one GPU, bucketed / single
# collective), rollout serving.
#   gpurun --timeout 2400 -- 'ROUND=r06 bash tools/evidence/workloads.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=${ROUND:-r06}; O=gpurun_out/$R; mkdir -p $O
for w in hlg_concept_d256 dhm_d256 mixed_d256; do
  timeout 300 python bench.py --workload $w --cpu-baseline off --steps 8 --warmup 3 --inclusive-pool > $O/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --workload hlg_ref --steps 256 --warmup 256 > $O/bench_hlg_ref.json 2>/dev/null
timeout 300 python bench.py --workload grid_ref --steps 100 --warmup 200 > $O/bench_grid_ref.json 2>/dev/null
timeout 200 python bench.py --minibatch 256 --cpu-baseline off --steps 64 --warmup 16 --inclusive-pool > $O/bench_hlg_d256_minibatch256.json 2>/dev/null
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off --inclusive-pool --no-ref-dims > $O/bench_rccl_single_rank.json 2> $O/rccl_single_rank.log
UPAMD_GRAD_BUCKETS=force timeout 300 python bench.py --gpus 2 --steps 12 --warmup 4 --minibatch 512 --inclusive-pool > $O/bench_2ranks_1gpu_gloo_bucketed.json 2> $O/r2b.err
UPAMD_GRAD_BUCKETS=0 timeout 300 python bench.py --gpus 2 --steps 12 --warmup 4 --minibatch 512 --inclusive-pool > $O/bench_2ranks_1gpu_gloo_single.json 2> $O/r2s.err
timeout 200 python tools/rollout_bench.py --D 16 --L 2 > $O/rollout_d16.json 2> $O/rollout_d16.err
timeout 200 python tools/rollout_bench.py --D 256 --L 3 --clients 8 16 32 64 --cpu-procs 1 16 --cpu-requests 10 > $O/rollout_d256.json 2> $O/rollout_d256.err
python tools/evidence/lines.py $O/bench_*.json
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/rollout_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['model'], d['cpu_select_action'])
        for s in d['serving']: print('   ', {k: (round(v,2) if isinstance(v,float) else v) for k,v in s.items()})
    except Exception as e: print(f, 'FAILED', e)
PY
