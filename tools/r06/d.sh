# round 6, call d: corrected new tests, default bench line (ref_dims) + self-spawned 2-rank lines, co-residency lab (b.sh),
# rollout measurements (c.sh), then a single-lane flake loop of the full suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06d; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_rollout.py "tests/test_gpu_update_branches.py::test_bucket_events_fire_only_behind_final_ranges" -x -q -m gpu 2>&1 | tail -40) > $O/new_tests.log 2>&1
tail -4 $O/new_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_gpus2_selfspawn.json 2> $O/bench_gpus2_selfspawn.err; echo "bench2 rc=$?"
timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --scaling strong > $O/bench_gpus2_strong_selfspawn.json 2> $O/bench_gpus2_strong.err; echo "bench2s rc=$?"
python tools/evidence/lines.py $O/bench_*.json
python - <<PY
import json
for f in ('bench_gpus2_selfspawn','bench_gpus2_strong_selfspawn'):
    try:
        d=json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1]); print(f, d.get('rccl_ranks_seen'), d.get('collective_backend'), d.get('allreduce_ms'), d.get('allreduce_buckets'), d['config']['global_batch'], d['scaling'])
    except Exception as e: print(f,'FAILED',e)
PY
bash tools/r06/b.sh
bash tools/r06/c.sh
AMD_LOG_LEVEL=1 bash tools/r06/flake_loop.sh 1 ${FLAKE_BUDGET:-1300}
