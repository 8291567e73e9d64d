# kernel timelines of the 256-row and the 2048-row step (idle gaps, concurrency, per-stream kernel time) -> gpurun_out/r03z/timeline_*.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
rm -rf /tmp/p_t1 /tmp/p_t2
rocprofv3 --kernel-trace -d /tmp/p_t1 -o tr -- python bench.py --minibatch 256 --cpu-baseline off --steps 40 --warmup 8 --no-kernel-events > /dev/null 2>&1
python tools/timeline_rocpd.py $(find /tmp/p_t1 -name "*.db" | head -1) $O/timeline_hlg_d256_minibatch256.txt 12
rocprofv3 --kernel-trace -d /tmp/p_t2 -o tr -- python bench.py --cpu-baseline off --steps 12 --warmup 3 --no-kernel-events > /dev/null 2>&1
python tools/timeline_rocpd.py $(find /tmp/p_t2 -name "*.db" | head -1) $O/timeline_hlg_d256.txt 4
head -50 $O/timeline_hlg_d256.txt
