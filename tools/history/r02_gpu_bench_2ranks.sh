# bench.py through the driver's multi-GPU launch line with 2 and 4 ranks SHARING the one GPU (gloo staging instead of RCCL): a functional
# check of the rank logic (same replay, one global permutation, sharded minibatches, barrier + max-over-ranks timing), not a scaling number
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
for n in 2 4; do
  UPAMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n \
     bench.py --gpus $n --steps 8 --warmup 2 --cpu-baseline off --minibatch 512 > $O/bench_${n}ranks_1gpu_gloo.json 2> $O/bench_${n}ranks_1gpu_gloo.err
  tail -1 $O/bench_${n}ranks_1gpu_gloo.json | cut -c1-400
  tail -3 $O/bench_${n}ranks_1gpu_gloo.err
done
UPAMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
     bench.py --gpus 2 --steps 8 --warmup 2 --cpu-baseline off --minibatch 1024 --scaling strong > $O/bench_2ranks_strong_1gpu_gloo.json 2> $O/bench_2ranks_strong_1gpu_gloo.err
tail -1 $O/bench_2ranks_strong_1gpu_gloo.json | cut -c1-400
