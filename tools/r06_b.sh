#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_encoder_fused.py tests/test_ops_gpu.py -x -q -m gpu -k "encoder or self_attention or fused or layer" 2>&1 | tail -4
echo "== trace"; GF_LIB_PATH=$PWD/tools/ab/k9p_trace.so timeout 300 python tools/k9p_trace.py 16 2>&1 | grep -v amdgpu.ids
timeout 900 python bench.py --no-cpu-baseline --no-train --steps 20 --warmup 5 > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err
tail -14 gpurun_out/r06b_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06b_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
print(json.dumps(d['side_measurements']['hpatches_b1'], indent=1))
PY
} > gpurun_out/r06_b.log 2>&1
cat gpurun_out/r06_b.log
