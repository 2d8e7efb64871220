cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train_ops.py -x -q -k "x3" > gpurun_out/r05_b_ops.log 2>&1; tail -25 gpurun_out/r05_b_ops.log
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -s -k "full_training_backward" > gpurun_out/r05_b_enc.log 2>&1; tail -12 gpurun_out/r05_b_enc.log | cut -c1-400
(timeout 600 python bench.py --mode train --dtype fp16x3 --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/r05_b_train_x3.json 2> gpurun_out/r05_b_train_x3.err; echo "train x3 rc=$?"; tail -c 300 gpurun_out/r05_b_train_x3.err)
(timeout 900 python bench.py > gpurun_out/r05_b_bench.json 2> gpurun_out/r05_b_bench.err; echo "bench all rc=$?"; tail -c 300 gpurun_out/r05_b_bench.err)
