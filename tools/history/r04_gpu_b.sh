# round 4, second GPU call: the fused small-model path -- parity, bit-identity, bench at the reference dims, host/GPU split
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_tiny.py -m gpu -x -q 2>&1 | tail -30) > $O/tests_tiny.log 2>&1
tail -5 $O/tests_tiny.log
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_update_branches.py tests/test_gpu_rollout.py tests/test_gpu_dp.py -m gpu -q 2>&1 | tail -30) > $O/tests_parity.log 2>&1
tail -5 $O/tests_parity.log
timeout 300 python tools/r04_diag_tiny.py hlg_ref > $O/diag_hlg_ref.log 2>&1
tail -25 $O/diag_hlg_ref.log | head -12
for t in 1024 512; do
  UPAMD_TUNE=tiny_threads=$t timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 32 --warmup 8 > $O/bench_hlg_ref_t$t.json 2>$O/bench_hlg_ref_t$t.err
  UPAMD_TUNE=tiny_threads=$t timeout 300 python bench.py --workload grid_ref --cpu-baseline off --steps 10 --warmup 4 > $O/bench_grid_ref_t$t.json 2>/dev/null
done
UPAMD_TUNE=tiny_fused=0 timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 32 --warmup 8 > $O/bench_hlg_ref_general.json 2>/dev/null
rm -rf /tmp/p_t1
rocprofv3 --kernel-trace --stats -d /tmp/p_t1 -o tr -- python bench.py --workload hlg_ref --cpu-baseline off --steps 32 --warmup 8 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_t1 -name "*.db" | head -1) $O/kernel_trace_hlg_ref_fused.txt
head -14 $O/kernel_trace_hlg_ref_fused.txt | cut -c1-60,100-170
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d.get('kernel_ms_per_step'))
except Exception as e:
    print('$f FAILED', e)
PY
done
