# round 5, very last call: the driver-style default line on the final tree (PMC-stamped fields filled)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05t; mkdir -p $O
timeout 230 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), round(d['ms_per_step'],3), round(r['frac'],4), r['traffic'], d['cpu_baseline']['value'], d['strong_proxy']['value'], d['strong_proxy']['value_with_route_overhead'], d['update_params_inclusive']['fraction_of_step_rate'], d['update_params_inclusive_records']['fraction_of_step_rate'])
PY
