# round 6, call f: the co-residency lab with the NT GEMM compiled for <= 128 VGPRs (tools/lab_census/build.py nt128): two of its
# workgroups now fit next to ONE resident message-passing workgroup (registers: 4 x 64 + 2 x 128 = 512 per SIMD lane; LDS: 80 + 2 x 32 KB)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; B=gpurun_out/r06f; mkdir -p $B
L=tools/lab_census/run.py
run() { # name mode tune census libdir [env]
  timeout 400 env $6 python $L --mode $2 --tune "$3" --census $4 --lib-dir $5 --out $B/$1.json > $B/$1.line 2> $B/$1.err || echo "$1 FAILED rc=$?"
  python - <<PY
import json
try:
    d=json.load(open('$B/$1.json')); c=d.get('census') or {}
    print('$1', d['library'], d.get('tune'), d.get('edge_lds_floor'), 'ms/round %.3f' % d['ms_per_round'], 'samples/s %.0f' % d['samples_per_s'], 'cu_share_both %.3f' % (c.get('cu_level') or {}).get('share', float('nan')))
except Exception as e: print('$1', 'no result', e)
PY
}
run n_serial serial "" 2 csrc_nt128
run n_serial_pad0 serial "gemm_lds_pad=0" 2 csrc_nt128
run n_serial_pad8k serial "gemm_lds_pad=8192" 2 csrc_nt128
run n_lanes lanes "" 2 csrc_nt128
run n_lanes_pad0 lanes "gemm_lds_pad=0" 2 csrc_nt128
run n_lanes_pad0_nostagger lanes "gemm_lds_pad=0,gemm_stagger_mode=0" 2 csrc_nt128
run n_lanes_pad0_floor lanes "gemm_lds_pad=0" 2 csrc_nt128 UPAMD_LAB_EDGE_LDS=97280
run n_wgrad2_pad0 wgrad2 "gemm_lds_pad=0" 2 csrc_nt128
run n_two_pad0 two_streams "gemm_lds_pad=0" 2 csrc_nt128
run o_serial serial "" 2 csrc
run nc_lanes_pad0 lanes "gemm_lds_pad=0" 1 csrc_nt128
run nc_lanes_pad0_floor lanes "gemm_lds_pad=0" 1 csrc_nt128 UPAMD_LAB_EDGE_LDS=97280
run nc_serial_pad0 serial "gemm_lds_pad=0" 1 csrc_nt128
# ---- the product's two-lane step (UPAMD_LANES=2) on the bench and under the census; its test
O=gpurun_out/r06f
(timeout 600 python -m pytest "tests/test_gpu_update_branches.py::test_two_lane_step_matches_the_one_lane_step" -x -q -m gpu 2>&1 | tail -15) > $O/lanes_test.log 2>&1; tail -2 $O/lanes_test.log
UPAMD_LANES=2 timeout 400 python bench.py --steps 20 --warmup 5 --cpu-baseline off --no-ref-dims --strong-proxy off > $O/bench_lanes2.json 2> $O/bench_lanes2.err
UPAMD_LANES=2 UPAMD_TUNE=gemm_lds_pad=0 timeout 400 python bench.py --steps 20 --warmup 5 --cpu-baseline off --no-ref-dims --strong-proxy off > $O/bench_lanes2_pad0.json 2> $O/bench_lanes2_pad0.err
python tools/evidence/lines.py $O/bench_*.json
run c_lanes lanes "" 1 csrc
run t_lanes lanes "" 0 csrc
AMD_LOG_LEVEL=1 bash tools/r06/flake_loop.sh 1 ${FLAKE_BUDGET:-1200}
