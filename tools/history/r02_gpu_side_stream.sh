# forked step (side stream): targeted tests + bench at 2048 / 256 rows / reference dims, both settings
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep_edge.py tests/test_gpu_mlp.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -12) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
for ss in 1 0; do
  UPAMD_TUNE=side_stream=$ss timeout 300 python bench.py --cpu-baseline off > $O/bench_default_ss$ss.json 2> $O/bench_default_ss$ss.err
  UPAMD_TUNE=side_stream=$ss timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/bench_mb256_ss$ss.json 2> $O/bench_mb256_ss$ss.err
  UPAMD_TUNE=side_stream=$ss timeout 300 python bench.py --workload hlg_ref --cpu-baseline off > $O/bench_hlg_ref_ss$ss.json 2> $O/bench_hlg_ref_ss$ss.err
done
for f in default_ss1 default_ss0 mb256_ss1 mb256_ss0 hlg_ref_ss1 hlg_ref_ss0; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3))
except Exception as e:
    print('$f', 'FAILED', e); print(open('$O/bench_$f.err').read()[-1500:])
PY
done
