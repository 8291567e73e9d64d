# round 6, call b: co-residency lab (review item 1).  Timing on the product library, then the census build: which CUs held GEMM and
# message-passing workgroups at the same time, and what that did to each.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06b; mkdir -p $O
L=tools/lab_census/run.py
run() { # name mode tune census [env]
  timeout 400 env $5 python $L --mode $2 --tune "$3" --census $4 --out $O/$1.json > $O/$1.line 2> $O/$1.err || echo "$1 FAILED rc=$?"
  python - <<PY
import json
try:
    d=json.load(open('$O/$1.json')); c=d.get('census') or {}
    print('$1', d['library'], 'ms/round %.3f' % d['ms_per_round'], 'samples/s %.0f' % d['samples_per_s'], 'cu_share_both %.3f' % (c.get('cu_level') or {}).get('share', float('nan')), 'chip both us %.0f' % (c.get('chip_us') or {}).get('both_in_flight', float('nan')))
except Exception as e: print('$1', 'no result', e)
PY
}
# ---- timing, product library
run t_serial serial "" 0
run t_wgrad2 wgrad2 "" 0
run t_wgrad2_pad0 wgrad2 "gemm_lds_pad=0" 0
run t_two two_streams "" 0
run t_two_pad0 two_streams "gemm_lds_pad=0" 0
run t_two_pad0_nostagger two_streams "gemm_lds_pad=0,gemm_stagger_mode=0" 0
# ---- timing, census build with the walk kernels held to ONE workgroup per CU (96 KB floor): 1 walk + 2 GEMM workgroups fit a CU
run t_serial_floor serial "" 2 UPAMD_LAB_EDGE_LDS=97280
run t_two_pad0_floor two_streams "gemm_lds_pad=0" 2 UPAMD_LAB_EDGE_LDS=97280
run t_wgrad2_pad0_floor wgrad2 "gemm_lds_pad=0" 2 UPAMD_LAB_EDGE_LDS=97280
# ---- census
run c_serial serial "" 1
run c_wgrad2 wgrad2 "" 1
run c_two two_streams "" 1
run c_two_pad0 two_streams "gemm_lds_pad=0" 1
run c_two_pad0_floor two_streams "gemm_lds_pad=0" 1 UPAMD_LAB_EDGE_LDS=97280
run c_wgrad2_pad0_floor wgrad2 "gemm_lds_pad=0" 1 UPAMD_LAB_EDGE_LDS=97280
tail -3 $O/*.err | tail -30
