#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest.log 2>&1
tail -5 gpurun_out/r06_gputest.log
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -12 gpurun_out/r06_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for k in d.get('roofline_kernels', []):
    print(f"  {k['tag']:22s} {k['avg_launch_ms']*1e3:8.1f} us x {k['launches_per_step']:5.1f} = {k['total_ms_per_step']:.3f} ms  frac {k['frac']:.3f}")
PY
