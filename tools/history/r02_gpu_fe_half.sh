# land-use head on the m half of FE alone (head.hip, tune knob fe_half): head tests + bench, both settings, + trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mlp.py -m gpu -q -x -k "wide_model or loss_and_gradients or mlp or forked" 2>&1 | tail -5) > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log
for fh in 1 0 1 0; do
  UPAMD_TUNE=fe_half=$fh timeout 300 python bench.py --cpu-baseline off --no-kernel-events | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fe_half=$fh', round(d['value']), round(d['ms_per_step'],3))"
done
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 --no-kernel-events > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
grep -E "head_|edge_fwd_kernelILb1|greduce" $O/kernel_trace_hlg_d256.txt | cut -c1-60,100-175
