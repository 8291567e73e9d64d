# round 6, call a: the new tests alone, the default bench line (ref_dims), bench.py --gpus 2 with no launcher (gloo, one GPU),
# then the flake loop (review item 2)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06a; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_rollout.py "tests/test_gpu_update_branches.py::test_bucket_events_fire_only_behind_final_ranges" -x -q -m gpu 2>&1 | tail -30) > $O/new_tests.log 2>&1
tail -5 $O/new_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_gpus2_selfspawn.json 2> $O/bench_gpus2_selfspawn.err; echo "bench2 rc=$?"
timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --scaling strong > $O/bench_gpus2_strong_selfspawn.json 2> $O/bench_gpus2_strong.err; echo "bench2s rc=$?"
python - <<PY
import json
for f in ('bench_default','bench_gpus2_selfspawn','bench_gpus2_strong_selfspawn'):
    try:
        d=json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],3), d.get('rccl_ranks_seen'), d.get('collective_backend'), d.get('allreduce_ms'), {k:(v.get('value'),v.get('ms_per_step'),(v.get('roofline') or {}).get('frac'), (v.get('cpu_baseline') or {}).get('value'), (v.get('update_params_inclusive') or {}).get('fraction_of_step_rate'), v.get('wall_s'), v.get('error')) for k,v in (d.get('ref_dims') or {}).items()})
    except Exception as e: print(f,'FAILED',e)
PY
bash tools/r06/flake_loop.sh 2 ${FLAKE_BUDGET:-2400}
