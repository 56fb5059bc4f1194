#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
GF_K9_PAIR=0 timeout 600 python tools/k9_pair_check.py save
GF_K9_PAIR=1 timeout 600 python tools/k9_pair_check.py save
timeout 300 python tools/k9_pair_check.py cmp | grep -v "^float16_2\|^bfloat16_2" 
echo "== tests (pair)"
GF_K9_PAIR=1 timeout 900 python -m pytest tests/test_encoder_fused.py -x -q -m gpu 2>&1 | tail -5
echo "== trace"; GF_LIB_PATH=$PWD/tools/ab/k9p_trace.so timeout 300 python tools/k9p_trace.py 16 2>&1 | grep -v amdgpu.ids
echo "== times"
for n in 16 8; do echo "pair images=$n"; timeout 300 python tools/k9_time.py $n 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r06_k9c.log 2>&1
rm -f gpurun_out/k9_pair_0.pt gpurun_out/k9_pair_1.pt
cat gpurun_out/r06_k9c.log
