#!/bin/bash
# r04: gloo rehearsals of bench.py --gpus 2 / 4 on the one GPU (functional: the N > 1 code path incl. the multi_gpu object; 4 ranks
# with the sharded forward issued eagerly and as graph segments), then the whole GPU suite five times back to back
O=${OUT_ROOT:-gpurun_out}/r04e; mkdir -p $O
PANDORA_DIST_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29502 bench.py --gpus 2 --steps 2 --warmup 1 --only 320x512 2> $O/bench_gloo_n2.err | grep '^{"metric"' > $O/bench_gloo_n2.json
PANDORA_DIST_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29504 bench.py --gpus 4 --steps 2 --warmup 1 --only 320x512 2> $O/bench_gloo_n4_eager.err | grep '^{"metric"' > $O/bench_gloo_n4_eager.json
PANDORA_DIST_BACKEND=gloo PANDORA_SEGMENT_GRAPHS=force timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29505 bench.py --gpus 4 --steps 2 --warmup 1 --only 320x512 2> $O/bench_gloo_n4_segments.err | grep '^{"metric"' > $O/bench_gloo_n4_segments.json
SUITE_DIR=r04e bash tools/diag/r04_suite.sh 5 run
