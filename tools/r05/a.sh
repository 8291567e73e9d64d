# round 5, call a: regression of the bucketed backward (full GPU suite), default bench line with the strong-scaling proxy,
# 256-row step under the bucket / stagger knobs, 2-rank functional lines (bucketed vs single collective)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05a; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25) > $O/gpu_tests.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline off > $O/bench_default.json 2> $O/bench_default.err
B="timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 96 --warmup 32 --strong-proxy off"
for t in "grad_buckets=1" "grad_buckets=0" "gemm_stagger_cycles=0" "gemm_stagger_cycles=12000" "gemm_stagger_cycles=24000" "gemm_stagger_mode=2" "gemm_stagger_mode=3" "side_wgrad=2" "side_wgrad=0"; do
  UPAMD_TUNE="$t" $B > $O/mb256_$t.json 2>/dev/null
done
R="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 12 --warmup 4 --minibatch 512"
UPAMD_DIST_BACKEND=gloo timeout 600 $R > $O/bench_2ranks_bucketed.json 2> $O/r2b.err
UPAMD_DIST_BACKEND=gloo UPAMD_GRAD_BUCKETS=0 timeout 600 $R > $O/bench_2ranks_single.json 2> $O/r2s.err
tail -6 $O/gpu_tests.log
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    k=d.get('kernel_ms_per_step') or {}
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round(r.get('frac') or 0,3), d.get('allreduce_ms'), (d.get('strong_proxy') or {}).get('value'), (d.get('strong_proxy') or {}).get('ms_share'), {a: round(b,3) for a,b in k.items()})
except Exception as e: print('$f', 'FAILED', e)
PY
done
