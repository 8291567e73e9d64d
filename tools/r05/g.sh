# round 5, call g: A/B of the walk prologue change (one LDS trip for row pointers + own rows; hoisted store base) against the
# previous build (tools/lab/libupamd_base.so), same box, alternating; then the parity tests that pin the message passing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05g; mkdir -p $O
B="timeout 200 python bench.py --steps 24 --warmup 6 --cpu-baseline off --strong-proxy off --inclusive-pool"
for i in 1 2 3; do
  UPAMD_LIB_PATH=$PWD/tools/lab/libupamd_base.so $B > $O/base_$i.json 2>/dev/null
  $B > $O/new_$i.json 2>/dev/null
done
UPAMD_LIB_PATH=$PWD/tools/lab/libupamd_base.so timeout 200 python bench.py --workload dhm_d256 --steps 8 --warmup 3 --cpu-baseline off --inclusive-pool > $O/base_dhm.json 2>/dev/null
timeout 200 python bench.py --workload dhm_d256 --steps 8 --warmup 3 --cpu-baseline off --inclusive-pool > $O/new_dhm.json 2>/dev/null
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print('%-14s %8d %8.4f  fwd %.3f bwd %.3f nt %.3f' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], k['edge_fwd'], k['edge_bwd'], k['gemm_nt_128']))
except Exception as e: print('$f', 'FAILED', e)
PY
done
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_deep_edge.py -m gpu -q -x 2>&1 | tail -6) > $O/tests.log 2>&1; tail -3 $O/tests.log
