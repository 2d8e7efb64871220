#!/bin/bash
# current breakdown of the training step: host issue time, per-class kernel time, device timeline
tag=${1:-r03}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python scripts/train_host_time.py > $out/${tag}_train_host_time.txt 2>&1
cd /tmp; rm -rf /tmp/proft_$tag
timeout 600 rocprofv3 --kernel-trace -d /tmp/proft_$tag -o train -- \
    python $repo/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/${tag}_train_rocprof.err
f=$(find /tmp/proft_$tag -name "*results.db" | head -1)
if [ -n "$f" ]; then
    python $repo/scripts/train_prof_summary.py $f 7 60 > $out/${tag}_train_kernel_stats.txt
    python $repo/scripts/train_timeline.py $f 5 >> $out/${tag}_train_kernel_stats.txt
fi
cd $repo
