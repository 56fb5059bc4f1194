#!/bin/bash
# round 6: wave-pair K9 against the four-wave K9 (same box): outputs, tests, times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
GF_K9_PAIR=0 timeout 600 python tools/k9_pair_check.py save
GF_K9_PAIR=1 timeout 600 python tools/k9_pair_check.py save
timeout 300 python tools/k9_pair_check.py cmp
echo "== tests (pair)"
GF_K9_PAIR=1 timeout 900 python -m pytest tests/test_encoder_fused.py -x -q -m gpu 2>&1 | tail -15
echo "== times"
for p in 0 1; do for n in 16 8; do echo "GF_K9_PAIR=$p images=$n"; GF_K9_PAIR=$p timeout 300 python tools/k9_time.py $n 2>&1 | grep -v amdgpu.ids; done; done
} > gpurun_out/r06_k9.log 2>&1
rm -f gpurun_out/k9_pair_0.pt gpurun_out/k9_pair_1.pt
tail -60 gpurun_out/r06_k9.log
