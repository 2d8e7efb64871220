cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_graph_train.py -q -x -k "replayed" > gpurun_out/r05_f_gt.log 2>&1; tail -12 gpurun_out/r05_f_gt.log | cut -c1-600
(timeout 900 python bench.py --mode train --steps 20 --warmup 8 --no-cpu-baseline > gpurun_out/r05_f_train.json 2> gpurun_out/r05_f_train.err; echo "train rc=$?"; tail -c 400 gpurun_out/r05_f_train.err)
(timeout 600 python bench.py --mode query --dtype bf16 --no-cpu-baseline > gpurun_out/r05_f_bf16.json 2> gpurun_out/r05_f_bf16.err; echo "bf16 rc=$?")
