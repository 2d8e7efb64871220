#!/bin/bash
# kernel durations of the query kernels at fit-loop and bench sizes, split kernel against the one-wave-per-head kernels
repo=$(pwd); export TMPDIR=/tmp; cd /tmp
for v in split nosplit; do
  if [ $v = nosplit ]; then export CHORE_QUERY_X3_NOSPLIT=1; else unset CHORE_QUERY_X3_NOSPLIT; fi
  rm -rf /tmp/qk_$v
  QT_MODES=fp16x3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/qk_$v -o q --output-format csv -- python $repo/scripts/query_time.py 1 6890 1 3000 8 6890 4 20000 > /dev/null 2>&1
  f=$(find /tmp/qk_$v -name "*kernel_trace.csv" | head -1)
  echo "== $v"; python $repo/scripts/prof_summary.py $f 40 | grep -i "query" | cut -c1-150
done
