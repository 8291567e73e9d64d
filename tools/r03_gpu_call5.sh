cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
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
run base "" ""
run sub2 "" "--sub-batches 2"
run sub2_e96_p0 "edge_min_lds=98304,gemm_lds_pad=0" "--sub-batches 2"
run sub2_e88_p0 "edge_min_lds=90112,gemm_lds_pad=0" "--sub-batches 2"
run sub2_e96_p12 "edge_min_lds=98304" "--sub-batches 2"
run sub2_e112_p0 "edge_min_lds=114688,gemm_lds_pad=0" "--sub-batches 2"
run sub1_e96 "edge_min_lds=98304" ""
