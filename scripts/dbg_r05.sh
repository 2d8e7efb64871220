cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --mode fit --no-cpu-baseline > gpurun_out/r05_i_fit.json 2> gpurun_out/r05_i_fit.err; echo rc=$?
timeout 900 python bench.py --mode fit --frames-per-gpu 8 --steps 2 --no-cpu-baseline > gpurun_out/r05_i_fit8.json 2> gpurun_out/r05_i_fit8.err; echo rc=$?
timeout 900 python -m pytest tests/test_gpu_fit_chain.py -q -x -k "pipelined or kept_graphs" 2>&1 | tail -3
