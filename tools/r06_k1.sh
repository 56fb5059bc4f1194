#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in base new1cu base new1cu; do
  echo "== $v"
  if [ $v = new1cu ]; then timeout 300 python tools/k1_time.py 8 0.2 2>&1 | grep "k1_"; else GF_LIB_PATH=$PWD/tools/ab/k1a_$v.so timeout 300 python tools/k1_time.py 8 0.2 2>&1 | grep "k1_"; fi
done
echo "== tests (new, 1 WG per CU)"
timeout 1200 python -m pytest tests/test_k1_dual_softmax_gpu.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r06_k1.log 2>&1
cat gpurun_out/r06_k1.log
