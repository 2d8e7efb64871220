#!/bin/bash
# per-kernel time of a traced fit chain and the launch sequence of one iteration of each phase
tag=${1:-r03}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/proff_$tag
timeout 600 rocprofv3 --kernel-trace -d /tmp/proff_$tag -o fit -- \
    python $repo/bench.py --mode fit --steps 1 --warmup 0 --no-cpu-baseline > $out/${tag}_fit_bench.json 2> $out/${tag}_fit_rocprof.err
f=$(find /tmp/proff_$tag -name "*results.db" | head -1)
if [ -n "$f" ]; then
    python $repo/scripts/train_prof_summary.py $f 1 50 > $out/${tag}_fit_kernel_stats.txt
    python $repo/scripts/fit_iter_trace.py $f 60 130 250 | cut -c1-160 >> $out/${tag}_fit_kernel_stats.txt
fi
cd $repo
