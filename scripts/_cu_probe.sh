mkdir -p gpurun_out
ok=0; bad=0
for i in $(seq 1 10); do
  timeout 120 python scripts/probes/graph_record_watchdog.py flat > /tmp/o.txt 2> /tmp/e.txt
  if [ $? = 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); grep -m1 "Error\|error" /tmp/e.txt | cut -c1-200; fi
done
echo "reproducer: ok=$ok bad=$bad"
for i in 1 2 3; do
  timeout 600 python bench.py > gpurun_out/r05_bench_$i.json 2> gpurun_out/r05_bench_$i.err; echo "bench $i rc=$?"
done
timeout 900 python -m pytest tests/test_gpu_graph_train.py tests/test_gpu_ddp_nccl.py tests/test_gpu_ddp.py tests/test_gpu_ddp_trainstep.py tests/test_gpu_fit_chain.py -x -q -m gpu 2>&1 | tail -3
