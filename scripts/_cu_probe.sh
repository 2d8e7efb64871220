mkdir -p gpurun_out
for w in 256 512; do
  CHORE_WGRAD128_WGS=$w timeout 300 python scripts/wgrad_layer_ab.py fp16x3 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wgs=$w', {k:v['us'] for k,v in d.items() if k.startswith('1x1')})"
done
for v in 0 1; do
  if [ $v = 1 ]; then export CHORE_WGRAD_NO128=1; fi
  timeout 300 python bench.py --mode train --steps 20 --warmup 8 --no-cpu-baseline > gpurun_out/train_128_$v.json 2> gpurun_out/train_128_$v.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/train_128_$v.json") if l.startswith("{")][-1])
print("no128=$v", round(d["ms_per_step"],3), "ms/step", d.get("allreduce",{}).get("ms_per_step_no_sync"))
PY
done
