cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6)
timeout 300 python bench.py --cpu-baseline off > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off > $O/bench_hlg_ref.json 2> $O/bench_hlg_ref.err
timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/bench_mb256.json 2>/dev/null
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
for f in default hlg_ref mb256; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), d.get('roofline',{}).get('achieved'))
except Exception as e:
    print('$f', 'FAILED', e); print(open('$O/bench_$f.err').read()[-1500:] if '$f' != 'mb256' else '')
PY
done
head -3 $O/kernel_trace_hlg_d256.txt; grep -E "pointer_bwd2|greduce|he_feat" $O/kernel_trace_hlg_d256.txt | cut -c1-150
