#!/bin/bash
# r04 third GPU call: probe, the whole GPU suite in ONE invocation with durations, attention shapes, bench A/Bs (CFG batch, parity)
O=${OUT_ROOT:-gpurun_out}/r04c; mkdir -p $O
python3 tools/diag/event_query_probe.py > $O/event_query_probe.txt 2>&1
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=60 > $O/full_suite.log 2>&1
echo "rc=$?" >> $O/full_suite.log
timeout 900 python3 tools/attn_shapes.py --rounds 7 > $O/attention_shapes.txt 2>&1
PANDORA_CFG_BATCH=1 timeout 600 python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --emulate-shard off > $O/bench_cfg_batch1.json 2> $O/bench_cfg_batch1.err
PANDORA_CFG_BATCH=0 timeout 600 python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --emulate-shard off > $O/bench_cfg_batch0.json 2> $O/bench_cfg_batch0.err
PANDORA_CFG_BATCH=1 timeout 600 python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --emulate-shard off > $O/bench_cfg_batch1_b.json 2> $O/bench_cfg_batch1_b.err
timeout 600 python3 bench.py --steps 10 --warmup 3 --parity --dtype f16 --cpu-baseline off --emulate-shard off > $O/bench_parity_f16.json 2> $O/bench_parity_f16.err
tail -n 4 $O/full_suite.log
