cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04t; mkdir -p $O
(UPAMD_TEST_TINY_THREADS=512 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/gpu_tests_512.log 2>&1
tail -4 $O/gpu_tests_512.log
