#!/bin/bash
# conv_pc_kernel<h16_t, 9, 8, 32, 3, 2> at 256^2 took ~1 s per launch in `bench.py --mode query --dtype fp16` under rocprofv3:
# reproduce with the shipped library and with a variant (CHORE_HIP_LIB) given as $1
repo=$(pwd); export TMPDIR=/tmp; cd /tmp
for lib in "" "$1"; do
  if [ -n "$lib" ]; then export CHORE_HIP_LIB=$repo/$lib; else unset CHORE_HIP_LIB; fi
  rm -rf /tmp/fp16q
  SECONDS=0; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fp16q -o q --output-format csv -- python $repo/bench.py --mode query --dtype fp16 --steps 4 --warmup 1 --no-cpu-baseline > /tmp/fp16q.json 2> /tmp/fp16q.err
  echo "wall $SECONDS s"; python -c "import json; d=json.load(open(\"/tmp/fp16q.json\")); print(\"ms_per_step\", d[\"ms_per_step\"])"
  f=$(find /tmp/fp16q -name "*kernel_trace.csv" | head -1)
  echo "== lib '${lib}'"; python $repo/scripts/prof_summary.py $f 4 | cut -c1-140
done
