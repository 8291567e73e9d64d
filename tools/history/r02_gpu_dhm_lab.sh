# large-graph size class of the forward with H left in HBM (tune knob fwd_h_hbm): targeted tests + dhm / mixed bench, both settings
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -m gpu -q -x -k "baseline_graph or dhm_graphs or wide_model or maximum_size or full_size or random_small" 2>&1 | tail -6) > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log
for hh in 1 0; do
  for w in dhm_d256 mixed_d256; do
    UPAMD_TUNE=fwd_h_hbm=$hh timeout 300 python bench.py --workload $w --cpu-baseline off > $O/bench_${w}_hh$hh.json 2> $O/bench_${w}_hh$hh.err
  done
done
for f in dhm_d256_hh1 dhm_d256_hh0 mixed_d256_hh1 mixed_d256_hh0; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'].get('edge_fwd'), d['kernel_ms_per_step'].get('edge_bwd'))
except Exception as e:
    print('$f', 'FAILED', e); print(open('$O/bench_$f.err').read()[-1500:])
PY
done
