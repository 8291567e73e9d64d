# round 5, call j: bucketed route after the two fixes (normal-priority communication stream, tail range issued from the caller's stream)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05j; mkdir -p $O
P=29580
run() { name=$1; rows=$2; shift 2; P=$((P+1))
  env UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$P "$@" timeout 200 python bench.py --minibatch $rows --steps 96 --warmup 32 --cpu-baseline off --inclusive-pool --no-kernel-events --strong-proxy off > $O/$name.json 2>> $O/err.log
}
for rep in 1 2 3; do
  run single_mb256_$rep 256 UPAMD_GRAD_BUCKETS=0
  run buckets_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force
  run buckets_tailcomm_mb256_$rep 256 UPAMD_GRAD_BUCKETS=force UPAMD_BUCKET_TAIL=comm
done
for rep in 1 2; do
  run single_mb2048_$rep 2048 UPAMD_GRAD_BUCKETS=0
  run buckets_mb2048_$rep 2048 UPAMD_GRAD_BUCKETS=force
done
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('%-30s %8d  %8.4f ms/step   host enqueue %7.4f' % ('$f'.split('/')[-1][:-5], d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))
except Exception as e: print('$f', 'FAILED', e)
PY
done; tail -2 $O/err.log
(timeout 300 python -m pytest tests/test_gpu_update_branches.py -m gpu -q -k "bucketed" 2>&1 | tail -3)
