# round 6, last check: prepare() chunks at the YAML dims (auto = 2 now) three times each, and the default line (fresh PMC stamps)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06j; mkdir -p $O
for i in 1 2 3; do
  timeout 200 python bench.py --workload hlg_ref --steps 64 --warmup 256 --cpu-baseline off > $O/hlg_ref_auto_$i.json 2>/dev/null
  UPAMD_PREPARE_CHUNKS=1 timeout 200 python bench.py --workload hlg_ref --steps 64 --warmup 256 --cpu-baseline off > $O/hlg_ref_chunks1_$i.json 2>/dev/null
done
(timeout 300 python -m pytest tests/test_gpu_update_branches.py tests/test_gpu_tiny.py -x -q -m gpu 2>&1 | tail -3) > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/hlg_ref_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); u=d['update_params_inclusive']; r=d['update_params_inclusive_records']
    print(f.split('/')[-1], round(d['ms_per_step'],4), 'tuples ms', round(1e3*u['seconds'],1), round(u['fraction_of_step_rate'],3), 'records ms', round(1e3*r['seconds'],1), round(r['fraction_of_step_rate'],3))
PY
python tools/evidence/lines.py $O/bench_default.json
