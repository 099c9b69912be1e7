#!/bin/bash
# the headline windows twice each way on ONE box: default streams, then AGX_NO_STREAM_PRIO=1 (tuning build)
for k in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default      ', round(d['value']), d['step_ms']['median'], d['kernel_only']['launch_ms'], '| phmm', round(d['pairhmm']['value']/1e6,1), d['pairhmm']['step_ms']['median'], d['pairhmm']['kernel_only']['launch_ms'])"
  AGX_NO_STREAM_PRIO=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no priorities', round(d['value']), d['step_ms']['median'], d['kernel_only']['launch_ms'], '| phmm', round(d['pairhmm']['value']/1e6,1), d['pairhmm']['step_ms']['median'], d['pairhmm']['kernel_only']['launch_ms'])"
done
