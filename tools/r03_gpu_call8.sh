cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -8) > $O/tests.log 2>&1
tail -5 $O/tests.log
run() { # name, tune, args
  UPAMD_TUNE=$2 timeout 300 python bench.py --cpu-baseline off $3 > $O/bench_$1.json 2> $O/bench_$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1])
    k=d.get('kernel_ms_per_step',{})
    print('$1', round(d['value']), round(d['ms_per_step'],3), {a:round(b,3) for a,b in k.items()})
except Exception as e:
    print('$1 FAILED', e); print(open('$O/bench_$1.err').read()[-800:])
PY
}
run hlg "" ""
run hlg2 "" ""
run dhm "" "--workload dhm_d256"
run dhm_fold2 fold_layer1=2 "--workload dhm_d256"
run mixed "" "--workload mixed_d256"
run mixed_fold2 fold_layer1=2 "--workload mixed_d256"
run mb256 "" "--minibatch 256 --steps 40 --warmup 8"
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 --no-kernel-events > /dev/null 2>&1
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
head -16 $O/kernel_trace_hlg_d256.txt | cut -c1-60,100-170
