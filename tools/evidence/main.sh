# Evidence pass 1 (GPU box; one gpurun call): GPU suite + smoke, the driver-style default bench line, rocprofv3 kernel traces of the
# same commands, the PMC passes behind roofline.traffic / valu_busy (stamped with the hash of csrc/).
#   gpurun --timeout 2400 -- 'ROUND=r06 bash tools/evidence/main.sh'      then here: bash tools/evidence/collect.sh r06
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=${ROUND:-r06}; O=gpurun_out/$R; mkdir -p $O
(timeout 1000 python -m pytest tests/ -x -q -m gpu --durations=8 2>&1 | tail -16) > $O/gpu_tests.log 2>&1
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
CMD="python bench.py --steps 4 --warmup 1 --cpu-baseline off --no-kernel-events --strong-proxy off --inclusive-pool --no-ref-dims"
rm -rf /tmp/p_tr /tmp/p_tr2 /tmp/p_tr3 /tmp/p_f /tmp/p_w /tmp/p_u
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 --strong-proxy off --inclusive-pool --no-ref-dims > $O/bench_default_under_rocprofv3.json 2> /dev/null
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/p_tr2 -o tr -- python bench.py --workload hlg_ref --cpu-baseline off --steps 128 --warmup 128 --inclusive-pool > $O/bench_hlg_ref_under_rocprofv3.json 2>/dev/null
python profiles/summarize_rocpd.py $(find /tmp/p_tr2 -name "*.db" | head -1) $O/kernel_trace_hlg_ref.txt
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/p_tr3 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events --inclusive-pool > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr3 -name "*.db" | head -1) $O/kernel_trace_hlg_d256_minibatch256.txt
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- $CMD > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- $CMD > $O/pmc_write.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/p_u -- $CMD > $O/pmc_util.log 2>&1
python tools/pmc_traffic.py /tmp/p_f /tmp/p_w --json $O/pmc_traffic.json --md $O/pmc_step_traffic.md --command "$CMD (hlg_d256, 1 x MI355X)" > /dev/null 2> $O/pmc_traffic.err
python tools/pmc_util.py /tmp/p_u --md $O/pmc_utilisation.md --json $O/pmc_util.json > /dev/null 2> $O/pmc_util.err
tail -5 $O/gpu_tests.log; tail -1 $O/smoke.log; head -c 300 $O/pmc_traffic.err; head -c 300 $O/pmc_util.err
python tools/evidence/lines.py $O/bench_*.json
