# round 4: weight-gradient placement at the strong-scaling operating points (side_wgrad 1 vs 2), with and without a process group;
# host threads A/B at D = 256
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04h; mkdir -p $O
for mb in 128 256 512; do
  for t in 1 2; do
    UPAMD_TUNE=side_wgrad=$t timeout 300 python bench.py --minibatch $mb --cpu-baseline off --steps 64 --warmup 16 > $O/bench_mb${mb}_w$t.json 2>/dev/null
    UPAMD_TUNE=side_wgrad=$t UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29532 timeout 300 python bench.py --minibatch $mb --cpu-baseline off --steps 64 --warmup 16 > $O/bench_mb${mb}_w${t}_rccl1.json 2>/dev/null
  done
done
timeout 300 python bench.py --cpu-baseline off --steps 24 --warmup 6 > $O/bench_default_a.json 2>/dev/null
OMP_NUM_THREADS=128 UPAMD_BENCH_THREADS=128 timeout 300 python bench.py --cpu-baseline off --steps 24 --warmup 6 > $O/bench_default_threads128.json 2>/dev/null
timeout 300 python bench.py --cpu-baseline off --steps 24 --warmup 6 > $O/bench_default_b.json 2>/dev/null
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round((d.get('roofline') or {}).get('frac') or 0,3), round(d['update_params_inclusive']['samples_per_s']))
except Exception as e: print('$f', 'FAILED', e)
PY
done
