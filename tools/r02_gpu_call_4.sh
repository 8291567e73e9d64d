cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for w in hlg_d256 hlg_ref; do
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace -d /tmp/prof_$w -o tr -- python bench.py --workload $w --cpu-baseline off --steps 20 --warmup 4 > gpurun_out/r02/prof_bench_$w.json 2> gpurun_out/r02/prof_bench_$w.err
  db=$(find /tmp/prof_$w -name "*.db" | head -1)
  python profiles/summarize_rocpd.py $db gpurun_out/r02/kernel_trace_$w.txt
  head -30 gpurun_out/r02/kernel_trace_$w.txt | cut -c1-200
done
