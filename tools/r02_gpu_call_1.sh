set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/r02/gpu_tests_1.log 2>&1
python __graft_entry__.py smoke > gpurun_out/r02/smoke_1.log 2>&1
timeout 600 python bench.py > gpurun_out/r02/bench_default_1.json 2> gpurun_out/r02/bench_default_1.err
RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 UPAMD_DIST_FORCE_INIT=1 NCCL_DEBUG=INFO timeout 300 python bench.py --steps 8 --warmup 2 --cpu-baseline off > gpurun_out/r02/rccl_force_init.log 2>&1
for w in dhm_d256 mixed_d256 hlg_ref; do timeout 600 python bench.py --workload $w --cpu-baseline off > gpurun_out/r02/bench_$w.json 2> gpurun_out/r02/bench_$w.err; done
tail -3 gpurun_out/r02/gpu_tests_1.log; cat gpurun_out/r02/smoke_1.log | tail -2; cut -c1-600 gpurun_out/r02/bench_default_1.json
