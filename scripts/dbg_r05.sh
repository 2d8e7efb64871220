cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/proft
timeout 600 rocprofv3 --kernel-trace -d /tmp/proft -o train -- python $GRAFT_REPO_ROOT/scripts/train_graph_trace.py 6 fp16x3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/tr.err
f=$(find /tmp/proft -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/train_step_trace.py $f $GRAFT_REPO_ROOT/gpurun_out/r05b_train_step_listing.txt > $GRAFT_REPO_ROOT/gpurun_out/r05b_train_kernel_stats.txt
head -30 $GRAFT_REPO_ROOT/gpurun_out/r05b_train_kernel_stats.txt | cut -c1-150
