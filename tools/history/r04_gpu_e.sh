cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04e; mkdir -p $O
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; python -c "import torch; print(torch.get_num_threads(), torch.__config__.parallel_info()[:400])"
OMP_NUM_THREADS=1 timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 256 --warmup 256 --no-kernel-events > $O/bench_omp1.json 2>/dev/null
timeout 300 python bench.py --workload hlg_ref --cpu-baseline off --steps 256 --warmup 256 --no-kernel-events > $O/bench_default.json 2>/dev/null
for f in $O/bench_omp1.json $O/bench_default.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['update_params_inclusive']['loop_s'])"; done
grep -i "throttled\|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null
