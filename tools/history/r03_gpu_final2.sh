# evidence of the round's LAST build (after the folded two-per-CU forms and the trimmed walk loops): GPU suite, smoke, PMC utilisation
# pass (feeds the bench line's message_passing.valu_busy), bench lines, kernel traces -> gpurun_out/r03z (copied to profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -22) > $O/gpu_tests.log 2>&1
python __graft_entry__.py smoke > $O/smoke.log 2>&1
CMD="python bench.py --steps 4 --warmup 1 --cpu-baseline off --no-kernel-events"
rm -rf /tmp/p_tr /tmp/p_tr3 /tmp/p_tr5 /tmp/p_u
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/p_u -- $CMD > $O/pmc_util.log 2>&1
python tools/pmc_util.py /tmp/p_u --md $O/pmc_utilisation.md --json $O/pmc_util.json > /dev/null && cp $O/pmc_util.json profiles/pmc_util.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --workload dhm_d256 --cpu-baseline off > $O/bench_dhm_d256.json 2>/dev/null
timeout 600 python bench.py --workload mixed_d256 --cpu-baseline off > $O/bench_mixed_d256.json 2>/dev/null
timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/bench_hlg_d256_minibatch256.json 2>/dev/null
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 timeout 600 python bench.py --steps 16 --warmup 4 --cpu-baseline off > $O/bench_rccl_single_rank.json 2>/dev/null
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29532 timeout 600 python bench.py --minibatch 256 --steps 40 --warmup 8 --cpu-baseline off > $O/bench_rccl_single_rank_minibatch256.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr3 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr3 -name "*.db" | head -1) $O/kernel_trace_hlg_d256_minibatch256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr5 -o tr -- python bench.py --workload dhm_d256 --cpu-baseline off --steps 6 --warmup 2 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr5 -name "*.db" | head -1) $O/kernel_trace_dhm_d256.txt
tail -3 $O/gpu_tests.log; tail -1 $O/smoke.log
for f in default dhm_d256 mixed_d256 hlg_d256_minibatch256 rccl_single_rank rccl_single_rank_minibatch256; do python - <<PY
import json
try:
    lines=open('$O/bench_$f.json').read().strip().splitlines()
    d=json.loads(lines[-1])
    cb=d.get('cpu_baseline') or {}
    print('$f', len(lines), round(d['value']), round(d['ms_per_step'],3), round(d.get('roofline',{}).get('achieved',0),1), round(d.get('roofline',{}).get('frac',0),3), 'incl', round(d['update_params_inclusive']['samples_per_s']), 'cpu', cb.get('value'), (d.get('message_passing') or {}).get('valu_busy'))
except Exception as e:
    print('$f FAILED', e)
PY
done
