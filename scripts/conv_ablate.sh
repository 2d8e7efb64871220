#!/bin/bash
# times the encoder under kernel ablation switches (results are numerically wrong when dbg != 0)
for d in 0 16 8; do
  echo "== CHORE_CONV_DBG=$d"
  CHORE_CONV_DBG=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('encode_ms', round(d['encode_ms'],3), 'conv3x3', round(sum(v['ms_per_step'] for n,v in k.items() if n.startswith('conv3x3')),3), 'conv1x1', round(k['conv1x1']['ms_per_step'],3))"
done
