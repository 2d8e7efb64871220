#!/bin/bash
# training-step and fit-loop evidence on the GPU box:  scripts/profile_train.sh r02
#   <tag>_bench_train.json / _bench_train_fp32.json / _bench_fit.json   bench.py lines of the other two modes
#   <tag>_train_kernel_stats.txt   per-kernel-class time of a traced training step (rocprofv3 --kernel-trace) + device timeline
#   <tag>_fit_kernel_stats.txt     the same for one traced fit_recon chain (300 iterations) + the launch sequence of one iteration of
#                                  each phase (optimize_smpl 'kpts', 'object only', 'joint')
tag=${1:-r02}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py --mode train --train-other-modes --steps 20 --warmup 5 > $out/${tag}_bench_train.json 2> $out/${tag}_bench_train.err
timeout 900 python bench.py --mode fit > $out/${tag}_bench_fit.json 2> $out/${tag}_bench_fit.err
# one REPLAYED training step (chore_amd.parallel.GraphedTrainStep): per-class kernel time, busy / idle, launch listing
cd /tmp; rm -rf /tmp/proft_$tag
timeout 600 rocprofv3 --kernel-trace -d /tmp/proft_$tag -o train -- \
    python $repo/scripts/train_graph_trace.py 6 fp16x3 > /dev/null 2> $out/${tag}_train_rocprof.err
f=$(find /tmp/proft_$tag -name "*results.db" | head -1)
if [ -n "$f" ]; then
    python $repo/scripts/train_step_trace.py $f $out/${tag}_train_step_listing.txt > $out/${tag}_train_kernel_stats.txt
fi
cd /tmp; rm -rf /tmp/proff_$tag
timeout 600 rocprofv3 --kernel-trace -d /tmp/proff_$tag -o fit -- \
    python $repo/bench.py --mode fit --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/${tag}_fit_rocprof.err
f=$(find /tmp/proff_$tag -name "*results.db" | head -1)
if [ -n "$f" ]; then
    python $repo/scripts/train_prof_summary.py $f 1 50 > $out/${tag}_fit_kernel_stats.txt
    python $repo/scripts/fit_iter_trace.py $f 60 130 250 | cut -c1-160 >> $out/${tag}_fit_kernel_stats.txt
fi
cd $repo
