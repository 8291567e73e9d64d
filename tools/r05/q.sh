# round 5, call q: hunt the intermittent failure of the bound-agent rollout test (full traceback this time)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05q; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_parity.py tests/test_gpu_rollout.py -m gpu -q -x --tb=long > $O/run0.log 2>&1; tail -3 $O/run0.log
for i in 1 2 3 4 5 6 7 8; do
  timeout 120 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x --tb=long > $O/run$i.log 2>&1
  tail -1 $O/run$i.log
done
grep -l "failed" $O/run*.log | head -3
f=$(grep -l "failed" $O/run*.log | head -1); if [ -n "$f" ]; then grep -n "Error\|^E " $f | head -30; fi
