# land-use head on the m half of FE alone (head.hip, tune knob fe_half): targeted tests + bench, both settings
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deep_edge.py tests/test_gpu_mlp.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -12) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
for fh in 1 0; do
  UPAMD_TUNE=fe_half=$fh timeout 300 python bench.py --cpu-baseline off > $O/bench_default_fh$fh.json 2> $O/bench_default_fh$fh.err
done
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 --no-kernel-events > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
for f in default_fh1 default_fh0; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'])
except Exception as e:
    print('$f', 'FAILED', e); print(open('$O/bench_$f.err').read()[-1500:])
PY
done
grep -E "head_|edge_fwd_kernelILb1|gemm_tn_mfma_kernelILi32|gemm_nt_mfma|he_bias" $O/kernel_trace_hlg_d256.txt | cut -c1-60,100-175
