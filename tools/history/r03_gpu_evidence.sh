# round-3 evidence run: GPU tests, smoke, bench lines (with CPU baselines for cfg-1 / cfg-2 / cfg-3), kernel traces, PMC passes,
# multi-rank functional runs on the one GPU -> gpurun_out/r03z (copied to profiles/ afterwards)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -30) > $O/gpu_tests.log 2>&1
python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --workload dhm_d256 --cpu-baseline quick > $O/bench_dhm_d256.json 2> $O/bench_dhm_d256.err
timeout 600 python bench.py --workload mixed_d256 --cpu-baseline off > $O/bench_mixed_d256.json 2> $O/bench_mixed_d256.err
timeout 900 python bench.py --workload hlg_ref --cpu-baseline full > $O/bench_hlg_ref.json 2> $O/bench_hlg_ref.err
timeout 900 python bench.py --workload grid_ref --cpu-baseline full > $O/bench_grid_ref.json 2> $O/bench_grid_ref.err
timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/bench_hlg_d256_minibatch256.json 2>/dev/null
timeout 900 python bench.py --cpu-baseline off --inclusive-unique > $O/bench_inclusive_unique.json 2> $O/bench_inclusive_unique.err
CMD="python bench.py --steps 4 --warmup 1 --cpu-baseline off --no-kernel-events"
rm -rf /tmp/p_tr /tmp/p_tr2 /tmp/p_tr3 /tmp/p_tr4 /tmp/p_f /tmp/p_w /tmp/p_u
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr2 -o tr -- python bench.py --workload hlg_ref --cpu-baseline off --steps 40 --warmup 8 > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr2 -name "*.db" | head -1) $O/kernel_trace_hlg_ref.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr3 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr3 -name "*.db" | head -1) $O/kernel_trace_hlg_d256_minibatch256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr4 -o tr -- python bench.py --workload mixed_d256 --cpu-baseline off --steps 8 --warmup 2 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr4 -name "*.db" | head -1) $O/kernel_trace_mixed_d256.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- $CMD > $O/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/p_u -- $CMD > $O/pmc_util.log 2>&1
python tools/pmc_traffic.py /tmp/p_f /tmp/p_w --json $O/pmc_traffic.json --md $O/pmc_step_traffic.md --command "$CMD (hlg_d256, 1 x MI355X)" > /dev/null
python tools/pmc_util.py /tmp/p_u --md $O/pmc_utilisation.md --json $O/pmc_util.json > /dev/null
# multi-rank functional runs: 2 / 4 ranks SHARING the one GPU over gloo (rank logic, not a scaling number); RCCL with one rank
for n in 2 4; do
  UPAMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n \
     bench.py --gpus $n --steps 8 --warmup 2 --cpu-baseline off --minibatch 512 > $O/bench_${n}ranks_1gpu_gloo_functional.json 2> $O/bench_${n}ranks_1gpu_gloo.err
done
UPAMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
     bench.py --gpus 2 --steps 8 --warmup 2 --cpu-baseline off --minibatch 1024 --scaling strong > $O/bench_2ranks_strong_1gpu_gloo_functional.json 2> $O/bench_2ranks_strong.err
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 NCCL_DEBUG=WARN timeout 600 python bench.py --steps 8 --warmup 2 --cpu-baseline off > $O/bench_rccl_single_rank.json 2> $O/rccl_force_init_single_rank.log
tail -3 $O/gpu_tests.log; tail -1 $O/smoke.log
for f in default dhm_d256 mixed_d256 hlg_ref grid_ref hlg_d256_minibatch256 inclusive_unique 2ranks_1gpu_gloo_functional 4ranks_1gpu_gloo_functional 2ranks_strong_1gpu_gloo_functional rccl_single_rank; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    cb=d.get('cpu_baseline') or {}
    print('$f', round(d['value']), round(d['ms_per_step'],3), round(d.get('roofline',{}).get('achieved',0),1), round(d.get('roofline',{}).get('frac',0),3), 'incl', round(d['update_params_inclusive']['samples_per_s']), 'cpu', cb.get('value'), cb.get('cores'), cb.get('one_thread'))
except Exception as e:
    print('$f FAILED', e)
PY
done
