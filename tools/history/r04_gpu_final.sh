# round 4, final evidence run on the frozen csrc/: GPU suite + smoke, driver-style default line, PMC passes (stamped with the csrc
# hash), kernel traces, the other workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04z; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14) > $O/gpu_tests.log 2>&1
python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
CMD="python bench.py --steps 4 --warmup 1 --cpu-baseline off --no-kernel-events"
rm -rf /tmp/p_tr /tmp/p_tr2 /tmp/p_tr3 /tmp/p_f /tmp/p_w /tmp/p_u
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 > $O/bench_default_under_rocprofv3.json 2> /dev/null
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr2 -o tr -- python bench.py --workload hlg_ref --cpu-baseline off --steps 128 --warmup 128 > $O/bench_hlg_ref_under_rocprofv3.json 2>/dev/null
python profiles/summarize_rocpd.py $(find /tmp/p_tr2 -name "*.db" | head -1) $O/kernel_trace_hlg_ref.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr3 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr3 -name "*.db" | head -1) $O/kernel_trace_hlg_d256_minibatch256.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- $CMD > $O/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/p_u -- $CMD > $O/pmc_util.log 2>&1
python tools/pmc_traffic.py /tmp/p_f /tmp/p_w --json $O/pmc_traffic.json --md $O/pmc_step_traffic.md --command "$CMD (hlg_d256, 1 x MI355X)" > /dev/null 2> $O/pmc_traffic.err
python tools/pmc_util.py /tmp/p_u --md $O/pmc_utilisation.md --json $O/pmc_util.json > /dev/null 2> $O/pmc_util.err
for w in hlg_concept_d256 dhm_d256 mixed_d256; do
  timeout 600 python bench.py --workload $w --cpu-baseline off --steps 8 --warmup 3 > $O/bench_$w.json 2>/dev/null
done
timeout 600 python bench.py --workload hlg_ref --steps 256 --warmup 256 > $O/bench_hlg_ref.json 2>/dev/null
timeout 600 python bench.py --workload grid_ref --steps 100 --warmup 200 > $O/bench_grid_ref.json 2>/dev/null
timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 64 --warmup 16 > $O/bench_hlg_d256_minibatch256.json 2>/dev/null
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline off > $O/bench_rccl_single_rank.json 2> $O/rccl_single_rank.log
tail -4 $O/gpu_tests.log; tail -1 $O/smoke.log; head -c 300 $O/pmc_traffic.err; head -c 300 $O/pmc_util.err
for f in $O/bench_*.json; do python - <<PY
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), round(r.get('frac') or 0,4), r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e: print('$f', 'FAILED', e)
PY
done
