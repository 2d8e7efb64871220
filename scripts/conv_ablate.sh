#!/bin/bash
# times the encoder under kernel ablation switches (results are numerically wrong when dbg != 0)
#   1 no weight loads  2 no patch loads  4 no MFMAs  8 no epilogue  16 no stat atomics  32 no patch publish  64 no GN prologue
for d in ${@:-0 16 8}; do
  echo "== CHORE_CONV_DBG=$d"
  CHORE_CONV_DBG=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernels']
print('encode_ms', round(d['encode_ms'],3), ' '.join('%s=%.3f' % (n.replace('conv_lds_kernel<unsigned short, ','c<'), v['ms_per_step']) for n,v in k.items() if n.startswith('conv')))"
done
