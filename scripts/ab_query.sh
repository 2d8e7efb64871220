#!/bin/bash
# same-box A/B of two builds of the library on the default query step (encode + query): scripts/ab_query.sh <other .so> [rounds] [extra bench args]
other=$1; n=${2:-3}; shift 2
for i in $(seq $n); do
  for v in new old; do
    if [ $v = old ]; then export CHORE_HIP_LIB=$PWD/$other; else unset CHORE_HIP_LIB; fi
    timeout 300 python bench.py --mode query --steps 30 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms_per_step', round(d['ms_per_step'],3), 'encode_ms', round(d.get('encode_ms',0),3), 'field_err', d['config'].get('field_err'))"
  done
done
