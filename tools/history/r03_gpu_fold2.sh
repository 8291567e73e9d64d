cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03w
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6) > $O/tests.log 2>&1
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
run dhm_new "" "--workload dhm_d256"
run dhm_old "fwd_h_hbm=0,bwd_nb_global=0" "--workload dhm_d256"
run dhm_fwdonly "bwd_nb_global=0" "--workload dhm_d256"
run dhm_new2 "" "--workload dhm_d256"
run mixed_new "" "--workload mixed_d256"
run mixed_old "fwd_h_hbm=0,bwd_nb_global=0" "--workload mixed_d256"
run mixed_new2 "" "--workload mixed_d256"
run hlg "" ""
