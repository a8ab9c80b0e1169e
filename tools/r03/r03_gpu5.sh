#!/bin/bash
OUT=gpurun_out/r03_gpu5; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "rccl or ripple" > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log; tail -n 15 $OUT/pytest.log
timeout -k 10 600 bash tools/scale_sweep.sh $OUT/sweep 3 2>&1 | tee $OUT/sweep.log | tail -n 20
REDSEC_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-sample 0 --no-mnist --gates 8192 > $OUT/rehearsal_2ranks.json 2> $OUT/rehearsal_2ranks.err; tail -c 700 $OUT/rehearsal_2ranks.json
