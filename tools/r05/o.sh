# round 5, call o: 4 ranks SHARING one GPU over gloo: single collective vs bucketed (host-blocking staged collectives x time-slicing)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05o; mkdir -p $O
R4="python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29573 bench.py --gpus 4 --steps 8 --warmup 3 --minibatch 512 --cpu-baseline off"
UPAMD_DIST_BACKEND=gloo UPAMD_GRAD_BUCKETS=0 timeout 300 $R4 > $O/bench_4ranks_single.json 2> $O/err1
UPAMD_DIST_BACKEND=gloo UPAMD_GRAD_BUCKETS=force timeout 300 $R4 > $O/bench_4ranks_bucketed.json 2> $O/err2
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), d.get('allreduce_ms'), len(d.get('allreduce_buckets') or []), d['host_enqueue_ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)
PY
done
