#!/bin/bash
# config 3 and the other PairHMM legs with read trains (default) and without (AGX_PHMM_NO_TRAINS=1, tuning build), same box
for k in 1 2; do
  for env in "" "AGX_PHMM_NO_TRAINS=1"; do
    env $env python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pairhmm']
c=d.get('corpus_10s',{}); c5=d.get('config5',{})
print('%-22s C3 %.1f M pairs/s (step %.4f ms, kernel %.4f ms, waves %d, useful %.3f) | corpus f32 %.1f M | C5 shard f64 %.2f M' % ('$env' or 'default (trains)', p['value']/1e6, p['step_ms']['median'], p['kernel_only']['launch_ms'], p['waves'], p['useful_cell_fraction'], c.get('f32_fma',{}).get('pairs_per_s',0)/1e6, c5.get('value',0)/1e6))"
  done
done
