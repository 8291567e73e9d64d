cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03y
mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward_stages or loss_and_gradients or wide_model or random_small or virtual_g or forked" 2>&1 | tail -4) > $O/tests_attn.log 2>&1
tail -3 $O/tests_attn.log
rm -rf /tmp/p_a
rocprofv3 --kernel-trace --stats -d /tmp/p_a -o tr -- python bench.py --cpu-baseline off --steps 12 --warmup 3 --no-kernel-events > $O/bench_attn.json 2>/dev/null
python profiles/summarize_rocpd.py $(find /tmp/p_a -name "*.db" | head -1) $O/kernel_trace_attn.txt
grep -n "attn_\|gemm_nt_dma2" $O/kernel_trace_attn.txt | cut -c1-40,95-170
