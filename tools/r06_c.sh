#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_encoder_fused.py -x -q -m gpu 2>&1 | tail -4
echo "== trace"; GF_LIB_PATH=$PWD/tools/ab/k9p_trace.so timeout 300 python tools/k9p_trace.py 16 2>&1 | grep -v amdgpu.ids | head -24
for n in 16 8; do timeout 300 python tools/k9_time.py $n 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r06_c.log 2>&1
cat gpurun_out/r06_c.log
