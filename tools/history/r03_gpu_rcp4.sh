# (lab script of the shared-reciprocal experiment: needs profiles/r03_lab_shared_reciprocal.diff applied -- the edge_rcp4 knob does not exist at HEAD)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03x
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -30) > $O/tests.log 2>&1
tail -5 $O/tests.log
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
run hlg_rcp4 "" ""
run hlg_rcp1 "edge_rcp4=0" ""
run hlg_rcp4b "" ""
run dhm_rcp4 "" "--workload dhm_d256"
run dhm_rcp1 "edge_rcp4=0" "--workload dhm_d256"
run mb256_rcp4 "" "--minibatch 256 --steps 40 --warmup 8"
run mb256_rcp1 "edge_rcp4=0" "--minibatch 256 --steps 40 --warmup 8"
