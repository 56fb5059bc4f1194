#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== trace"; GF_LIB_PATH=$PWD/tools/ab/k9p_trace.so timeout 300 python tools/k9p_trace.py 16 2>&1 | grep -v amdgpu.ids
for v in ; do echo "== $v"; GF_LIB_PATH=$PWD/tools/ab/k9p_$v.so timeout 300 python tools/k9_time.py 16 2>&1 | grep "enc_layer"; done
echo "== default"; timeout 300 python tools/k9_time.py 16 2>&1 | grep "enc_layer"
} > gpurun_out/r06_k9b.log 2>&1
cat gpurun_out/r06_k9b.log
