#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_outcome_parity_gpu.py tests/test_e2e_gpu.py -x -q -s -m gpu -k "hpatches_protocol or 640_16bit" > gpurun_out/r06_d.log 2>&1
grep -v "^$\|amdgpu.ids" gpurun_out/r06_d.log | tail -60
