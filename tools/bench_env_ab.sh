#!/bin/bash
# usage: tools/bench_env_ab.sh "<label>=<ENV=V ...>" ...   (runs bench.py --cpu-baseline off per setting, prints ms/step)
for spec in "$@"; do
  label="${spec%%=*}"; envs="${spec#*=}"
  out=$(env $envs python bench.py --steps 10 --warmup 3 --cpu-baseline off 2>/dev/null)
  echo "$label: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('320x512 %.2f ms  576x1024 %.2f ms' % (d['ms_per_step'], d['res_576x1024']['ms_per_step']))")"
done
