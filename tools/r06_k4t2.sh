#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_hip_backward.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06_k4t_tests2.log
timeout 900 python tools/train_attn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_train_attn_ab.txt
