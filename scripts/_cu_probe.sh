for f in "" "1,256,256,128:4256;1,128,256,128:4256"; do
  CHORE_PC_FORCE="$f" timeout 300 python scripts/conv_layer_ab.py fp16x3 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('force=$f', {k:v['us'] for k,v in d.items() if k.startswith('1x1')})"
done
CHORE_PC_FORCE="1,256,256,128:4256;1,128,256,128:4256" timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_config2.py -x -q -m gpu 2>&1 | tail -3
