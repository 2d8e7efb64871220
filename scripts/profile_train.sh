#!/bin/bash
# training-step profile on the GPU box:  scripts/profile_train.sh r01
tag=${1:-r01}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python bench.py --mode train --steps 20 --warmup 5 > $out/${tag}_bench_train.json 2> $out/${tag}_bench_train.err
python bench.py --mode train --steps 20 --warmup 5 --dtype fp32 > $out/${tag}_bench_train_fp32.json 2>> $out/${tag}_bench_train.err
cd /tmp; rm -rf /tmp/proft_$tag
rocprofv3 --kernel-trace -d /tmp/proft_$tag -o $tag --output-format csv -- \
    python $repo/bench.py --mode train --steps 16 --warmup 4 > /dev/null 2> $out/${tag}_train_rocprof.err
f=$(find /tmp/proft_$tag -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $repo/scripts/train_prof_summary.py $f > $out/${tag}_train_kernel_stats.txt
cd $repo
