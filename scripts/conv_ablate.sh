#!/bin/bash
# times the encoder under kernel ablation switches (results are numerically wrong when dbg != 0)
for d in 0 1 2 3 4 8 12 15; do
  echo "== CHORE_CONV_DBG=$d"
  CHORE_CONV_DBG=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('encode_ms', round(d['encode_ms'],3), 'n128', round(k['conv3x3_n128']['ms_per_step'],3), 'n64', round(k['conv3x3_n64']['ms_per_step'],3))"
done
