# round-3 call 1: the new parity tests + a baseline bench line of the unchanged kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_update_branches.py tests/test_gpu_rollout.py -m gpu -x -q -s 2>&1 | tail -60) > $O/new_tests.log 2>&1
timeout 600 python bench.py --cpu-baseline off > $O/bench_default.json 2> $O/bench_default.err
tail -40 $O/new_tests.log; cut -c1-400 $O/bench_default.json
