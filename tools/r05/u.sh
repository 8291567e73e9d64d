# round 5: the bound-agent rollout test BEHIND the heavy tests of the suite (the context in which it had failed), full traceback kept
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05u; mkdir -p $O
timeout 215 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_rollout.py -m gpu -q -x --tb=long > $O/run.log 2>&1; tail -3 $O/run.log; grep -n "^E \|retrying" $O/run.log | head
