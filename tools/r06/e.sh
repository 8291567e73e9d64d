# round 6, call e: census (co-residency evidence), the product's two-lane step on the bench, its test, rollout serving in the steady
# state, then the single-lane flake loop again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06e; mkdir -p $O
(timeout 900 python -m pytest "tests/test_gpu_update_branches.py::test_two_lane_step_matches_the_one_lane_step" tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -30) > $O/new_tests.log 2>&1; tail -3 $O/new_tests.log
L=tools/lab_census/run.py; B=gpurun_out/r06b; mkdir -p $B
run() { # name mode tune census [env]
  timeout 400 env $5 python $L --mode $2 --tune "$3" --census $4 --out $B/$1.json > $B/$1.line 2> $B/$1.err || echo "$1 FAILED rc=$?"
  python - <<PY
import json
try:
    d=json.load(open('$B/$1.json')); c=d.get('census') or {}
    print('$1', d['library'], 'ms/round %.3f' % d['ms_per_round'], 'samples/s %.0f' % d['samples_per_s'], 'cu_share_both %.3f' % (c.get('cu_level') or {}).get('share', float('nan')), 'chip both us %.0f' % (c.get('chip_us') or {}).get('both_in_flight', float('nan')))
except Exception as e: print('$1', 'no result', e)
PY
}
run t_lanes lanes "" 0
run t_lanes_pad0 lanes "gemm_lds_pad=0" 0
run t_serial_censuslib serial "" 2
run t_serial_floor serial "" 2 UPAMD_LAB_EDGE_LDS=97280
run t_two_pad0_floor two_streams "gemm_lds_pad=0" 2 UPAMD_LAB_EDGE_LDS=97280
run t_wgrad2_pad0_floor wgrad2 "gemm_lds_pad=0" 2 UPAMD_LAB_EDGE_LDS=97280
run t_lanes_pad0_floor lanes "gemm_lds_pad=0" 2 UPAMD_LAB_EDGE_LDS=97280
run c_serial serial "" 1
run c_wgrad2 wgrad2 "" 1
run c_two two_streams "" 1
run c_lanes lanes "" 1
run c_two_pad0 two_streams "gemm_lds_pad=0" 1
run c_two_pad0_floor two_streams "gemm_lds_pad=0" 1 UPAMD_LAB_EDGE_LDS=97280
run c_wgrad2_pad0_floor wgrad2 "gemm_lds_pad=0" 1 UPAMD_LAB_EDGE_LDS=97280
UPAMD_LANES=2 timeout 400 python bench.py --steps 20 --warmup 5 --cpu-baseline off --no-ref-dims --strong-proxy off > $O/bench_lanes2.json 2> $O/bench_lanes2.err
python tools/evidence/lines.py $O/bench_*.json
bash tools/r06/c.sh
AMD_LOG_LEVEL=1 bash tools/r06/flake_loop.sh 1 ${FLAKE_BUDGET:-1500}
