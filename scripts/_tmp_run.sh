cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/proft
timeout 600 rocprofv3 --kernel-trace -d /tmp/proft -o train -- python $GRAFT_REPO_ROOT/scripts/train_graph_trace.py 6 fp16x3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/tr.err
f=$(find /tmp/proft -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/train_step_trace.py $f $GRAFT_REPO_ROOT/gpurun_out/r05c_train_step_listing.txt > $GRAFT_REPO_ROOT/gpurun_out/r05c_train_kernel_stats.txt
head -40 $GRAFT_REPO_ROOT/gpurun_out/r05c_train_kernel_stats.txt | cut -c1-130
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --mode train --train-other-modes --steps 20 --warmup 8 --no-cpu-baseline > gpurun_out/r05_m_train.json 2> gpurun_out/r05_m_train.err; python -c "import json;d=json.load(open('gpurun_out/r05_m_train.json'));print(d['ms_per_step'], {m:v['ms_per_step'] for m,v in d['other_modes'].items()}, d['allreduce']['variants'])"
