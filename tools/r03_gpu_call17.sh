cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03n
mkdir -p $O
run() { # name, tune, env-prefix
  env $3 UPAMD_TUNE=$2 timeout 300 python bench.py --cpu-baseline off > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    k=d.get('kernel_ms_per_step',{})
    print('$1', round(d['value']), round(d['ms_per_step'],3), d.get('dp_mode'), {a:round(b,3) for a,b in k.items()})
except Exception as e:
    print('$1 FAILED', e); print(open('$O/bench_$1.err').read()[-800:])
PY
}
F="UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541"
run plain "" "A=1"
run rccl "" "$F"
run rccl_heads0 "side_heads=0,side_wgrad=0" "$F"
run gloo "" "$F UPAMD_DIST_BACKEND=gloo"
run plain2 "" "A=1"
