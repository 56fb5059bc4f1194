echo "== head-pipelined (default build)" > gpurun_out/r05r_k9.log
python tools/k9_digest.py 16 >> gpurun_out/r05r_k9.log 2>&1
python tools/k9_digest.py 8 2>&1 | tail -1 >> gpurun_out/r05r_k9.log
python -m pytest tests/test_encoder_fused.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/r05r_k9.log
HIPCC_EXTRA="-DK9_HEADPIPE=0" python -m geoformer_amd.build -q >> gpurun_out/r05r_k9.log 2>&1
echo "== round-4 phases (-DK9_HEADPIPE=0, GF_K9_HEADPIPE=0)" >> gpurun_out/r05r_k9.log
GF_K9_HEADPIPE=0 python tools/k9_digest.py 16 >> gpurun_out/r05r_k9.log 2>&1
GF_K9_HEADPIPE=0 python tools/k9_digest.py 8 2>&1 | tail -1 >> gpurun_out/r05r_k9.log
grep -v amdgpu.ids gpurun_out/r05r_k9.log
