cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03s
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forked or virtual or update_params or loss_and_gradients or wide_model_matches" 2>&1 | tail -4) > $O/tests.log 2>&1
tail -3 $O/tests.log
run() { # name, tune, args, env
  env $4 UPAMD_TUNE=$2 timeout 300 python bench.py --cpu-baseline off $3 > $O/bench_$1.json 2> $O/bench_$1.err
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
F="UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541"
A="--minibatch 256 --steps 40 --warmup 8"
run mb256 "" "$A" "X=1"
run mb256_nowg "side_wgrad=0" "$A" "X=1"
run mb256_rccl "" "$A" "$F"
run mb256_rccl_nowg "side_wgrad=0" "$A" "$F"
run big "" "" "X=1"
run big_rccl "" "" "$F"
