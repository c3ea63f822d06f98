#!/bin/bash
# r04 (final tree): gloo rehearsals of bench.py --gpus 2 / 4 on the one GPU - the N > 1 code path incl. the multi_gpu object
O=${OUT_ROOT:-gpurun_out}/r04k; mkdir -p $O
for n in 2 4; do
  PANDORA_DIST_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 2 --warmup 1 --only 320x512 > $O/bench_gloo_n$n.json 2> $O/bench_gloo_n$n.err
  echo "n=$n rc=$?" >> $O/rc.txt
done
