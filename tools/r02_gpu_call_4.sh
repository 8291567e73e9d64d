cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for w in hlg_d256 hlg_ref; do
  timeout 600 python bench.py --workload $w --cpu-baseline off --steps 24 --warmup 8 > gpurun_out/r02/bench_fused_$w.json 2> gpurun_out/r02/bench_fused_$w.err
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace -d /tmp/prof_$w -o tr -- python bench.py --workload $w --cpu-baseline off --steps 20 --warmup 4 > gpurun_out/r02/prof_bench_$w.json 2> gpurun_out/r02/prof_bench_$w.err
  db=$(find /tmp/prof_$w -name "*.db" | head -1)
  python profiles/summarize_rocpd.py $db gpurun_out/r02/kernel_trace_fused_$w.txt
  head -34 gpurun_out/r02/kernel_trace_fused_$w.txt | cut -c1-180
done
timeout 300 python bench.py --workload hlg_d256 --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > gpurun_out/r02/bench_fused_mb256.json 2>/dev/null
python - <<'PY'
import json
for f in ('bench_fused_hlg_d256','bench_fused_hlg_ref','bench_fused_mb256'):
    d=json.loads(open('gpurun_out/r02/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['value']), d['ms_per_step'], d.get('roofline',{}).get('achieved'), d['update_params_inclusive']['samples_per_s'])
PY
