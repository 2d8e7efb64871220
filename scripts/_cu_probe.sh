mkdir -p gpurun_out
CHORE_BENCH_RECORDS_DEADLINE_S=15 timeout 300 python bench.py > gpurun_out/bench_dog.json 2> gpurun_out/bench_dog.err; echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_dog.json") if l.startswith("{")][-1])
print("dog:", d["value"], d.get("records_aborted"), [k for k in ("train","fit","fit_fp16_fields") if k in d])
PY
date +%s > /tmp/t0
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "rc=$?"
echo "seconds: $(( $(date +%s) - $(cat /tmp/t0) ))"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_full.json") if l.startswith("{")][-1])
print("full:", d["value"], d["ms_per_step"], d.get("records_aborted"), d["train"]["value"], d["train"]["ms_per_step"], d["fit"]["value"], d["fit"]["loader_loop"]["pipelined"]["steady_state_ms_per_frame"], d["fit_fp16_fields"]["value"])
PY
