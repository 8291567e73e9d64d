cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05s; mkdir -p $O
for i in 1 2; do timeout 100 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x --tb=long > $O/run$i.log 2>&1; tail -1 $O/run$i.log; done
grep -h "^E " $O/run*.log | head -10
