cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q -k "gemm_tn or gemm_nt_mfma" 2>&1 | tail -8) > $O/tests_gemm.log 2>&1
tail -5 $O/tests_gemm.log
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide_model_matches or saturating or dma or update_params" 2>&1 | tail -8) > $O/tests_par.log 2>&1
tail -5 $O/tests_par.log
run() { # name, env, args
  UPAMD_TUNE=$2 timeout 300 python bench.py --cpu-baseline off $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    k=d.get('kernel_ms_per_step',{})
    print('$1', round(d['value']), round(d['ms_per_step'],3), 'nt', round(d['roofline']['achieved'],1), {a:round(b,3) for a,b in k.items()})
except Exception as e:
    print('$1 FAILED', e); print(open('$O/bench_$1.err').read()[-800:])
PY
}
run tn1 gemm_tn_dma=1 ""
run tn0 gemm_tn_dma=0 ""
run tn1b gemm_tn_dma=1 ""
run mb256 "" "--minibatch 256 --steps 40 --warmup 8"
run mb256_nostagger gemm_stagger_mode=0 "--minibatch 256 --steps 40 --warmup 8"
run mb256_stag12k gemm_stagger_cycles=12000 "--minibatch 256 --steps 40 --warmup 8"
run big_nostagger gemm_stagger_mode=0 ""
rm -rf /tmp/p_tr /tmp/p_tr3
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --workload dhm_d256 --cpu-baseline off --steps 8 --warmup 2 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_dhm_d256.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_tr3 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr3 -name "*.db" | head -1) $O/kernel_trace_mb256.txt
head -30 $O/kernel_trace_dhm_d256.txt
