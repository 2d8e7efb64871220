#!/bin/bash
# same-box A/B of two builds of the library on the training step: scripts/ab_train.sh <other .so> [rounds]
other=$1; n=${2:-3}
for i in $(seq $n); do
  for v in new old; do
    if [ $v = old ]; then export CHORE_HIP_LIB=$PWD/$other; else unset CHORE_HIP_LIB; fi
    timeout 200 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3))"
  done
done
