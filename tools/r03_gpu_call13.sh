cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mlp.py tests/test_gpu_deep_edge.py -m gpu -x -q 2>&1 | tail -6) > $O/tests.log 2>&1
tail -4 $O/tests.log
run() { # name, tune, args
  UPAMD_TUNE=$2 timeout 300 python bench.py --cpu-baseline off $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    k=d.get('kernel_ms_per_step',{})
    print('$1', round(d['value']), round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items()})
except Exception as e:
    print('$1 FAILED', e); print(open('$O/bench_$1.err').read()[-800:])
PY
}
run new "" ""
run old "side_heads=0,side_wgrad=0" ""
run new_b "" ""
run old_b "side_heads=0,side_wgrad=0" ""
run mb256_new "" "--minibatch 256 --steps 40 --warmup 8"
run mb256_old "side_heads=0,side_wgrad=0" "--minibatch 256 --steps 40 --warmup 8"
run ref_new "" "--workload hlg_ref --steps 40 --warmup 8"
run ref_old "side_heads=0,side_wgrad=0" "--workload hlg_ref --steps 40 --warmup 8"
