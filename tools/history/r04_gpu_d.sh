cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04d; mkdir -p $O
for t in 512 1024; do
UPAMD_TUNE=tiny_threads=$t timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 256 --warmup 256 > $O/bench_hlg_ref_t${t}.json 2>/dev/null
UPAMD_TUNE=tiny_threads=$t timeout 300 python bench.py --workload grid_ref --cpu-baseline off --steps 100 --warmup 200 > $O/bench_grid_ref_t${t}.json 2>/dev/null
done
UPAMD_TUNE=tiny_fused=0 timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 256 --warmup 256 > $O/bench_hlg_ref_general.json 2>/dev/null
UPAMD_TUNE=tiny_fused=0 timeout 300 python bench.py --workload grid_ref --cpu-baseline off --steps 100 --warmup 200 > $O/bench_grid_ref_general.json 2>/dev/null
timeout 300 python bench.py --cpu-baseline off --steps 20 --warmup 5 > $O/bench_default.json 2>/dev/null
timeout 300 python bench.py --cpu-baseline off --minibatch 256 --steps 64 --warmup 32 > $O/bench_mb256.json 2>/dev/null
for f in $O/bench_*.json; do python - <<PY
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round(d['update_params_inclusive']['loop_s'],4), d['update_params_inclusive']['optimizer_steps'])
PY
done
