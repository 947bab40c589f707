cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6f; D=/tmp/d8; mkdir -p $D
export PDA_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 PDA_BENCH_DUMP=$D
C="--workload c3 --steps 2 --warmup 1 --no-train --no-cpu-baseline --no-per-config --eval-block 262144"
( time timeout 300 python bench.py $C --extras-path $D/one.json > gpurun_out/r6f/one.out 2> gpurun_out/r6f/one.err ) 2>&1 | grep real
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 $C --extras-path $D/eight.json > gpurun_out/r6f/eight.out 2> gpurun_out/r6f/eight.err ) 2>&1 | grep real
tail -3 gpurun_out/r6f/eight.out | cut -c1-600; tail -30 gpurun_out/r6f/eight.err | cut -c1-300
ls $D
