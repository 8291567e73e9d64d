# round 5, call c: the new GPU tests, the inclusive-call breakdown under the packer changes (thread counts), rollout serving bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05c; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_update_branches.py tests/test_gpu_tiny.py -m gpu -q --durations=5 2>&1 | tail -25) > $O/gpu_tests.log 2>&1
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>&1
for t in 0 8 16 32 64; do
  UPAMD_PACK_THREADS=$t timeout 600 python tools/inclusive_breakdown.py --workload hlg_ref --unique > $O/breakdown_hlg_ref_threads$t.json 2>> $O/breakdown.err
done
for t in 0 16 32; do
  UPAMD_PACK_THREADS=$t timeout 600 python tools/inclusive_breakdown.py --workload hlg_d256 --unique > $O/breakdown_hlg_d256_threads$t.json 2>> $O/breakdown.err
done
timeout 900 python tools/rollout_bench.py --D 16 --L 2 > $O/rollout_d16.json 2> $O/rollout_d16.err
timeout 900 python tools/rollout_bench.py --D 256 --L 3 --clients 16 64 --cpu-procs 1 16 --cpu-requests 10 > $O/rollout_d256.json 2> $O/rollout_d256.err
timeout 900 python tools/rollout_bench.py --D 16 --L 2 --tuples --clients 16 64 --cpu-procs 1 > $O/rollout_d16_tuples.json 2> $O/rollout_d16_tuples.err
tail -8 $O/gpu_tests.log; cat $O/host.txt; tail -n 3 $O/breakdown.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/breakdown_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], {k: round(v,1) for k,v in d.items() if k.startswith('call_')}, {k: round(v,1) for k,v in d['phases_ms'].items()})
    except Exception as e: print(f, 'FAILED', e)
for f in sorted(glob.glob('$O/rollout_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['model'], d['state'], d['cpu_select_action'])
        for s in d['serving']: print('   ', {k: (round(v,2) if isinstance(v,float) else v) for k,v in s.items()})
    except Exception as e: print(f, 'FAILED', e)
PY
tail -n 5 $O/rollout_*.err
