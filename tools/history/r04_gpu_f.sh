# round 4: full suite on the current build, HBM stream yardstick, isolated vs concurrent attention / head kernels, 2-rank line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04f; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -15) > $O/gpu_tests.log 2>&1
python tools/hbm_peak.py > $O/hbm_peak.txt 2>&1
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 256 --warmup 256 > $O/bench_hlg_ref.json 2>/dev/null
UPAMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 6 --warmup 2 --minibatch 512 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err
for t in default side_heads=0 side_stream=0; do
  rm -rf /tmp/p_tr
  if [ $t = default ]; then E=""; else E="UPAMD_TUNE=$t"; fi
  env $E rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 > $O/prof_bench_$t.json 2> /dev/null
  python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256_$t.txt
done
tail -3 $O/gpu_tests.log; cat $O/hbm_peak.txt
python - <<PY
import json
for f in ('bench_hlg_ref','bench_2ranks_gloo','prof_bench_default','prof_bench_side_heads=0','prof_bench_side_stream=0'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],4), d.get('roofline',{}).get('frac'), d.get('allreduce_ms'))
    except Exception as e: print(f,'FAILED',e)
PY
