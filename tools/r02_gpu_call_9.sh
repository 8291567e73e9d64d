cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "wide_model" 2>&1 | tail -6)
for cfg in "gemm_split=0" "gemm_split=6"; do
UPAMD_TUNE=$cfg timeout 300 python bench.py --cpu-baseline off > $O/b_$cfg.json 2> $O/b.err
python - <<PY
import json
try:
    d=json.loads(open('$O/b_$cfg.json').read().strip().splitlines()[-1])
    k=d['kernel_ms_per_step']
    print('$cfg', round(d['value']), round(d['ms_per_step'],3), d['roofline']['achieved'], {a:round(b,3) for a,b in k.items()})
except Exception as e:
    print('$cfg FAILED', e); print(open('$O/b.err').read()[-800:])
PY
done
