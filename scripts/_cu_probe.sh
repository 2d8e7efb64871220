mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fit_chain.py tests/test_gpu_fit_stress.py -x -q -m gpu > gpurun_out/fit_tests.txt 2>&1
tail -15 gpurun_out/fit_tests.txt
for v in 0 1; do
  if [ $v = 1 ]; then export CHORE_FIT_SPLIT_RULE=1; fi
  timeout 300 python bench.py --mode fit --steps 3 --warmup 1 > gpurun_out/fit_rule_$v.json 2> gpurun_out/fit_rule_$v.err
  python - <<PY
import json
for l in open("gpurun_out/fit_rule_$v.json"):
    if l.startswith("{"):
        d=json.loads(l); ll=d["loader_loop"]; print("split=$v", "iter ms", round(d["value"],4), "chain", round(d["chain_ms_median"],1), {k:round(v["median_ms_per_iter"],4) for k,v in d["per_phase"].items()}, "serial", round(ll["serial"]["steady_state_ms_per_frame"],1), "pipelined", round(ll["pipelined"]["steady_state_ms_per_frame"],1), ll["pipelined"]["gaps_ms"])
PY
done
