cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hub or dma or forward_stages or loss_and_gradients or full_size or wide_model_matches" 2>&1 | tail -8) > $O/tests_hub.log 2>&1
tail -5 $O/tests_hub.log
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
run hub0 edge_hub_thr=0 ""
run hub16 edge_hub_thr=16 ""
run hub8 edge_hub_thr=8 ""
run hub12 edge_hub_thr=12 ""
run hub24 edge_hub_thr=24 ""
run hub0b edge_hub_thr=0 ""
run dhm_hub16 "" "--workload dhm_d256"
run mixed_hub16 "" "--workload mixed_d256"
run mb256_hub16 "" "--minibatch 256 --steps 40 --warmup 8"
run mb256_hub0 edge_hub_thr=0 "--minibatch 256 --steps 40 --warmup 8"
