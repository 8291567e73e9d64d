cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -60) > $O/gpu_tests.log 2>&1
tail -45 $O/gpu_tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
