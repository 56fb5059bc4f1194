#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_hip_backward.py tests/test_train_gpu.py -x -q -m gpu -s -k "batched_geo or train_step or training_step" 2>&1 | grep -i "image by image\|cosine\|passed\|failed\|error" | tee gpurun_out/r06_k4t_tests2.log
timeout 900 python tools/train_attn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_train_attn_ab.txt
