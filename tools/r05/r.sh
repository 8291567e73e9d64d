# round 5, call r: rollout tests after the worker-exit hardening
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05r; mkdir -p $O
for i in 1 2 3; do timeout 120 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x --tb=long > $O/run$i.log 2>&1; tail -1 $O/run$i.log; done
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
