# round-2 evidence run (session 2): GPU tests, smoke, bench lines, kernel traces, PMC passes -> gpurun_out/r02z (copied to profiles/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02z
mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/gpu_tests.log 2>&1
python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in dhm_d256 mixed_d256 hlg_ref; do timeout 600 python bench.py --workload $w --cpu-baseline off > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/bench_hlg_d256_minibatch256.json 2>/dev/null
UPAMD_TUNE=gemm_split=6 timeout 300 python bench.py --cpu-baseline off > $O/bench_optin_gemm_split6.json 2>/dev/null
CMD="python bench.py --steps 4 --warmup 1 --cpu-baseline off --no-kernel-events"
rm -rf /tmp/p_tr /tmp/p_tr2 /tmp/p_tr3 /tmp/p_f /tmp/p_w /tmp/p_u
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr2 -o tr -- python bench.py --workload hlg_ref --cpu-baseline off --steps 40 --warmup 8 > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr2 -name "*.db" | head -1) $O/kernel_trace_hlg_ref.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr3 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr3 -name "*.db" | head -1) $O/kernel_trace_hlg_d256_minibatch256.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- $CMD > $O/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/p_u -- $CMD > $O/pmc_util.log 2>&1
python tools/pmc_traffic.py /tmp/p_f /tmp/p_w --json $O/pmc_traffic.json --md $O/pmc_step_traffic.md --command "$CMD (hlg_d256, 1 x MI355X)" > /dev/null
python tools/pmc_util.py /tmp/p_u --md $O/pmc_utilisation.md > /dev/null
tail -3 $O/gpu_tests.log; tail -1 $O/smoke.log; cut -c1-300 $O/bench_default.json
for f in default dhm_d256 mixed_d256 hlg_ref hlg_d256_minibatch256 optin_gemm_split6; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['frac'],3))
except Exception as e:
    print('$f FAILED', e)
PY
done
