# round-3 call 2: LDS-DMA stage-in -- parity tests of the new paths, then A/B bench lines (pq_exp on / off), dhm / mixed
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dma or saturating or wide_model_matches or forward_stages or loss_and_gradients" --durations=8 2>&1 | tail -40) > $O/tests_dma.log 2>&1
tail -25 $O/tests_dma.log
for v in 1 0; do
  UPAMD_TUNE=pq_exp=$v timeout 300 python bench.py --cpu-baseline off > $O/bench_pqexp$v.json 2> $O/bench_pqexp$v.err
done
for w in dhm_d256 mixed_d256; do timeout 300 python bench.py --workload $w --cpu-baseline off > $O/bench_$w.json 2> $O/bench_$w.err; done
UPAMD_TUNE=bwd_nb_global=0 timeout 300 python bench.py --workload dhm_d256 --cpu-baseline off > $O/bench_dhm_nbg0.json 2>/dev/null
for f in pqexp1 pqexp0 dhm_d256 mixed_d256 dhm_nbg0; do python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), d.get('kernel_ms_per_step'))
except Exception as e:
    print('$f FAILED', e); print(open('$O/bench_$f.err').read()[-1500:] if '$f'!='dhm_nbg0' else '')
PY
done
