#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "wgrad or hip_conv" 2>&1 | tail -8 | tee gpurun_out/r06_wgrad_tests.log
timeout 600 python tools/k10_wgrad_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_wgrad_time.txt
