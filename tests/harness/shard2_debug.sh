mkdir -p gpurun_out/r3a
export PAG_BENCH_SINGLE_DEVICE=1
SZ="--reads 3000 --ref-len 3000000 --steps 1 --warmup 1 --no-cpu-baseline"
PAG_WALK_DEBUG=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --mode shard $SZ > gpurun_out/r3a/dbg2.out 2> gpurun_out/r3a/dbg2.err
echo "rc=$?"
grep -c "job done\|segment" gpurun_out/r3a/dbg2.err
tail -5 gpurun_out/r3a/dbg2.err
PAG_WALK_WAVES_PER_CU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 2 --mode shard $SZ > gpurun_out/r3a/w1.out 2> gpurun_out/r3a/w1.err
echo "rc=$?"; tail -3 gpurun_out/r3a/w1.err; tail -2 gpurun_out/r3a/w1.out
