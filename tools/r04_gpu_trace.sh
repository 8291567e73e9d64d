cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04t; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_tiny.py -m gpu -q -x 2>&1 | tail -8) > $O/gpu_tests_tiny.log 2>&1
tail -8 $O/gpu_tests_tiny.log
