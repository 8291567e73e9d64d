# round 4, first GPU call: full GPU suite on the pruned build + binding changes, baseline bench lines, 256-row knob sweep, kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25) > $O/gpu_tests.log 2>&1
python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
B256="python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8"
timeout 300 $B256 > $O/bench_mb256.json 2>/dev/null
for t in side_wgrad=0 side_wgrad=2 side_wgrad=3 side_priority=0 side_heads=0; do
  UPAMD_TUNE=$t timeout 300 $B256 > $O/bench_mb256_$t.json 2>/dev/null
done
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29532 timeout 300 $B256 > $O/bench_mb256_rccl1.json 2>$O/bench_mb256_rccl1.err
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 200 --warmup 20 > $O/bench_hlg_ref.json 2>/dev/null
timeout 300 python bench.py --workload grid_ref --cpu-baseline off --steps 200 --warmup 20 > $O/bench_grid_ref.json 2>/dev/null
rm -rf /tmp/p_tr /tmp/p_tr3
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr3 -o tr -- python bench.py --workload hlg_ref --cpu-baseline off --steps 50 --warmup 8 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr3 -name "*.db" | head -1) $O/kernel_trace_hlg_ref.txt
tail -4 $O/gpu_tests.log; tail -1 $O/smoke.log
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), round((d.get('roofline') or {}).get('frac') or 0,3))
except Exception as e:
    print('$f FAILED', e)
PY
done
