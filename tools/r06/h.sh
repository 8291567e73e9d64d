# round 6, call h: the default bench line on the final tree (fresh PMC summaries, measured route overhead), then the flake loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06h; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python tools/evidence/lines.py $O/bench_default.json
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); p=d['strong_proxy']
print({k:p[k] for k in ('ms_full','ms_share','value','value_with_route_overhead','rccl_route_overhead_ms')}); print(d['roofline'].get('traffic'), (d.get('message_passing') or {}).get('valu_busy'))
PY
AMD_LOG_LEVEL=1 bash tools/r06/flake_loop.sh 1 ${FLAKE_BUDGET:-3000}
cat gpurun_out/flake/serving_stats.jsonl 2>/dev/null | tail -20
ls gpurun_out/flake/abort_* 2>/dev/null
