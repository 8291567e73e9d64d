# round 5, call h: what the bucketed all-reduce costs under a REAL (one-rank) RCCL group: single collective vs four bucketed
# collectives vs the same four issued late vs two; 256 and 2048 rows; host enqueue time per step next to the step time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05h; mkdir -p $O
P=29540
run() { # name rows env...
  name=$1; rows=$2; shift 2; P=$((P+1))
  env UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$P "$@" timeout 200 python bench.py --minibatch $rows --steps 96 --warmup 32 --cpu-baseline off --inclusive-pool --no-kernel-events --strong-proxy off > $O/$name.json 2>> $O/err.log
}
for rep in 1 2; do
  run nogroup_mb256_$rep 256 UPAMD_DIST_FORCE_INIT=0
  run single_mb256_$rep 256 UPAMD_GRAD_BUCKETS=0
  run buckets_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force
  run late_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force UPAMD_BUCKET_LAB=late
  run two_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force UPAMD_BUCKET_LAB=two
done
run single_mb2048 2048 UPAMD_GRAD_BUCKETS=0
run buckets_mb2048 2048 UPAMD_GRAD_BUCKETS=force
run late_mb2048 2048 UPAMD_GRAD_BUCKETS=force UPAMD_BUCKET_LAB=late
run two_mb2048 2048 UPAMD_GRAD_BUCKETS=force UPAMD_BUCKET_LAB=two
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('%-22s %8d  %8.4f ms/step   host enqueue %7.4f ms/step' % ('$f'.split('/')[-1][:-5], d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))
except Exception as e: print('$f', 'FAILED', e)
PY
done; tail -2 $O/err.log
