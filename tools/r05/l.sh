# round 5, call l: the per-engine knob test + the suites that go through NativeEngine most (after the tuned() bracket went in)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05l; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_tiny.py tests/test_gpu_step_kernels.py tests/test_gpu_parity.py tests/test_gpu_rollout.py -m gpu -q -x 2>&1 | tail -8) > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 200 python bench.py --workload hlg_ref --steps 256 --warmup 256 --cpu-baseline off --inclusive-pool 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hlg_ref', round(d['value']), d['ms_per_step'], d['host_enqueue_ms_per_step'])"
