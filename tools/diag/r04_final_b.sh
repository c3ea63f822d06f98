#!/bin/bash
# r04: gloo rehearsals of bench.py --gpus 2 / 4 on the one GPU (functional: the N > 1 code path incl. the multi_gpu object),
# then the round's measurement artefacts (tools/collect_profiles_r04.sh)
O=${OUT_ROOT:-gpurun_out}/r04e; mkdir -p $O
for n in 2 4; do
  PANDORA_DIST_BACKEND=gloo PANDORA_SEGMENT_GRAPHS=force timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 2 --warmup 1 --only 320x512 > $O/bench_gloo_n$n.json 2> $O/bench_gloo_n$n.err
done
bash tools/collect_profiles_r04.sh all
