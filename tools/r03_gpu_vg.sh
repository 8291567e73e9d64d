cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -12) > $O/tests.log 2>&1
tail -8 $O/tests.log
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
run vg1 virtual_g=1 ""
run vg0 virtual_g=0 ""
run vg1b virtual_g=1 ""
run vg0b virtual_g=0 ""
run mb256_vg1 virtual_g=1 "--minibatch 256 --steps 40 --warmup 8"
run mb256_vg0 virtual_g=0 "--minibatch 256 --steps 40 --warmup 8"
