#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_hip_backward.py -x -q -m gpu -k "full_attention" 2>&1 | tail -15 | tee gpurun_out/r06_k4t_tests.log
timeout 300 python tools/k4_train_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_k4t_time.txt
