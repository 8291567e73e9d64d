cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
true
rm -rf /tmp/p_tr
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 > $O/prof_bench.json 2> $O/prof_bench.err
python profiles/summarize_rocpd.py $(find /tmp/p_tr -name "*.db" | head -1) $O/kernel_trace_mb256.txt
grep -E "chain_|gtn|greduce|gsmm|permute|total kernel" $O/kernel_trace_mb256.txt | cut -c1-150
