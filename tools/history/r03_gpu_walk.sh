cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_stages or loss_and_gradients or wide_model or saturating or dma_stage_in" 2>&1 | tail -8) > $O/tests.log 2>&1
tail -3 $O/tests.log
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
run hlg "" ""
run hlg2 "" ""
run dhm "" "--workload dhm_d256"
run mixed "" "--workload mixed_d256"
run mb256 "" "--minibatch 256 --steps 40 --warmup 8"
