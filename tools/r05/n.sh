# round 5, call n: functional multi-rank lines on ONE GPU over gloo (rank logic, bucketed route with 4 ranks; not scaling numbers)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05n; mkdir -p $O
R4="python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 4 --steps 8 --warmup 3 --minibatch 512 --cpu-baseline off"
UPAMD_DIST_BACKEND=gloo timeout 400 $R4 > $O/bench_4ranks_1gpu_gloo_bucketed.json 2> $O/r4b.err
R2S="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --steps 12 --warmup 4 --minibatch 1024 --scaling strong --cpu-baseline off"
UPAMD_DIST_BACKEND=gloo timeout 300 $R2S > $O/bench_2ranks_strong_1gpu_gloo_bucketed.json 2> $O/r2s.err
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), d['n_gpus'], d['scaling'], d.get('allreduce_ms'), d.get('allreduce_buckets'), d['dp_mode'])
except Exception as e: print('$f', 'FAILED', e)
PY
done; tail -3 $O/r4b.err
