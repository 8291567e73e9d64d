# round 2, session 2, call 3: edge_bwd one-rcp packed walk, greduce loads in flight, attention loads before the mask branch, lin_rows vector path for N >= 128 only
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
timeout 300 python bench.py --cpu-baseline off > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off > $O/bench_hlg_ref.json 2> $O/bench_hlg_ref.err
timeout 300 python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/bench_mb256.json 2> $O/bench_mb256.err
rm -rf /tmp/p_tr2
rocprofv3 --kernel-trace --stats -d /tmp/p_tr2 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events > $O/prof_mb256.json 2> $O/prof_mb256.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr2 -name "*.db" | head -1) $O/kernel_trace_mb256.txt
for f in default hlg_ref mb256; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), d.get('roofline',{}).get('achieved'))
except Exception as e:
    print('$f', 'FAILED', e); print(open('$O/bench_$f.err').read()[-1500:])
PY
done
grep -E "chain_|greduce|attn" $O/kernel_trace_mb256.txt | cut -c1-60,100-175
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --cpu-baseline off --steps 20 --warmup 4 --no-kernel-events > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_hlg_d256.txt
cut -c1-60,100-175 $O/kernel_trace_hlg_d256.txt | head -24
