# round 5, call i: the fixed cost of the bucketed route under a one-rank RCCL group -- the communication stream's priority level
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05i; mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > $O/prio.txt 2>&1; cat $O/prio.txt
P=29560
run() { name=$1; rows=$2; shift 2; P=$((P+1))
  env UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$P "$@" timeout 200 python bench.py --minibatch $rows --steps 96 --warmup 32 --cpu-baseline off --inclusive-pool --no-kernel-events --strong-proxy off > $O/$name.json 2>> $O/err.log
}
for rep in 1 2; do
  run single_mb256_$rep 256 UPAMD_GRAD_BUCKETS=0
  run buckets_prio-1_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force UPAMD_COMM_PRIORITY=-1
  run buckets_prio0_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force UPAMD_COMM_PRIORITY=0
  run buckets_prio1_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force UPAMD_COMM_PRIORITY=1
  run buckets_prio0_sideprio0_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force UPAMD_COMM_PRIORITY=0 UPAMD_TUNE=side_priority=0
done
run single_mb2048 2048 UPAMD_GRAD_BUCKETS=0
run buckets_prio0_mb2048 2048 UPAMD_GRAD_BUCKETS=force UPAMD_COMM_PRIORITY=0
run buckets_prio1_mb2048 2048 UPAMD_GRAD_BUCKETS=force UPAMD_COMM_PRIORITY=1
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('%-36s %8d  %8.4f ms/step   host enqueue %7.4f' % ('$f'.split('/')[-1][:-5], d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))
except Exception as e: print('$f', 'FAILED', e)
PY
done; tail -2 $O/err.log
