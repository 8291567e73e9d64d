# last check of the round's final build: the whole GPU suite, smoke, the default bench line, the same under a one-rank RCCL group
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03x
mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/gpu_tests.log 2>&1
python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
UPAMD_DIST_FORCE_INIT=1 RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 timeout 600 python bench.py --steps 16 --warmup 4 --cpu-baseline off > $O/bench_rccl_single_rank.json 2> $O/rccl.err
UPAMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29502 bench.py --gpus 2 --steps 8 --warmup 2 --cpu-baseline off --minibatch 512 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err
tail -3 $O/gpu_tests.log; tail -1 $O/smoke.log
for f in default rccl_single_rank 2ranks_gloo; do python - <<PY
import json
try:
    lines=open('$O/bench_$f.json').read().strip().splitlines()
    assert len(lines)==1, 'stdout has %d lines' % len(lines)
    d=json.loads(lines[-1])
    print('$f', round(d['value']), round(d['ms_per_step'],3), round(d.get('roofline',{}).get('achieved',0),1), d.get('dp_mode'), 'incl', round(d['update_params_inclusive']['samples_per_s']))
except Exception as e:
    print('$f FAILED', e)
PY
done
