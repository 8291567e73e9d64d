cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04t; mkdir -p $O
timeout 300 python tools/lab_trace/trace_tiny.py hlg_ref > $O/trace_hlg_ref.log 2>&1
head -3 $O/trace_hlg_ref.log | cut -c1-200
grep -A16 "by source line" $O/trace_hlg_ref.log | cut -c1-150
