# Co-residency lab of round 6 (profiles/r06_lab_coresidency.md): every configuration timed on the product
# library (census 0), on the census build without recording (census 2: its lab knobs UPAMD_LAB_EDGE_LDS / nt128 included) and with one
# step recorded (census 1).  Builds first, here:  python tools/lab_census/build.py ; python tools/lab_census/build.py nt128
#   gpurun --timeout 2400 -- 'bash tools/r06/lab_coresidency.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; B=gpurun_out/r06_lab; mkdir -p $B
L=tools/lab_census/run.py
run() { # name mode tune census libdir [env]
  timeout 400 env $6 python $L --mode $2 --tune "$3" --census $4 --lib-dir $5 --out $B/$1.json > $B/$1.line 2> $B/$1.err || echo "$1 FAILED rc=$?"
  python - <<PY
import json
try:
    d=json.load(open('$B/$1.json')); c=d.get('census') or {}
    print('$1', d['library'], d.get('tune'), d.get('edge_lds_floor'), 'ms/round %.3f' % d['ms_per_round'], 'cu_share_both %.3f' % (c.get('cu_level') or {}).get('share', float('nan')))
except Exception as e: print('$1', 'no result', e)
PY
}
FLOOR=UPAMD_LAB_EDGE_LDS=97280
# ---- step time, product library
run t_serial serial "" 0 csrc
run t_wgrad2 wgrad2 "" 0 csrc
run t_wgrad2_pad0 wgrad2 "gemm_lds_pad=0" 0 csrc
run t_two two_streams "" 0 csrc
run t_two_pad0 two_streams "gemm_lds_pad=0" 0 csrc
run t_two_pad0_nostagger two_streams "gemm_lds_pad=0,gemm_stagger_mode=0" 0 csrc
run t_lanes lanes "" 0 csrc
# ---- step time, census build (recording off): walks held to one workgroup per CU; the <= 128-VGPR NT GEMM
run t_serial_censuslib serial "" 2 csrc
run t_serial_floor serial "" 2 csrc $FLOOR
run t_two_pad0_floor two_streams "gemm_lds_pad=0" 2 csrc $FLOOR
run t_wgrad2_pad0_floor wgrad2 "gemm_lds_pad=0" 2 csrc $FLOOR
run n_serial serial "" 2 csrc_nt128
run n_serial_pad0 serial "gemm_lds_pad=0" 2 csrc_nt128
run n_lanes_pad0 lanes "gemm_lds_pad=0" 2 csrc_nt128
run n_lanes_pad0_floor lanes "gemm_lds_pad=0" 2 csrc_nt128 $FLOOR
run n_wgrad2_pad0 wgrad2 "gemm_lds_pad=0" 2 csrc_nt128
run n_two_pad0 two_streams "gemm_lds_pad=0" 2 csrc_nt128
# ---- census
run c_serial serial "" 1 csrc
run c_wgrad2 wgrad2 "" 1 csrc
run c_two two_streams "" 1 csrc
run c_lanes lanes "" 1 csrc
run c_two_pad0 two_streams "gemm_lds_pad=0" 1 csrc
run c_two_pad0_floor two_streams "gemm_lds_pad=0" 1 csrc $FLOOR
run c_wgrad2_pad0_floor wgrad2 "gemm_lds_pad=0" 1 csrc $FLOOR
run nc_serial_pad0 serial "gemm_lds_pad=0" 1 csrc_nt128
run nc_lanes_pad0 lanes "gemm_lds_pad=0" 1 csrc_nt128
run nc_lanes_pad0_floor lanes "gemm_lds_pad=0" 1 csrc_nt128 $FLOOR
