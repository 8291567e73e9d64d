cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
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
run wg0 side_wgrad=0 ""
run wg1 side_wgrad=1 ""
run wg0b side_wgrad=0 ""
run wg1b side_wgrad=1 ""
run wg1_pad0 "side_wgrad=1,gemm_lds_pad=0" ""
run mb256_wg1 side_wgrad=1 "--minibatch 256 --steps 40 --warmup 8"
run mb256_wg0 side_wgrad=0 "--minibatch 256 --steps 40 --warmup 8"
