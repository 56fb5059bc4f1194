python tools/k4_ab.py 1195 1,8 1,8,nopre 1,8 1,8,nopre > gpurun_out/r05q_k4ab.log 2>&1
python tools/k4_ab.py 333 1,8 1,8,nopre >> gpurun_out/r05q_k4ab.log 2>&1
python tools/k4_ab.py 70 1,8 1,8,nopre >> gpurun_out/r05q_k4ab.log 2>&1
python -m pytest tests/test_ops_gpu.py -k self_attention -q -m gpu 2>&1 | tail -2 >> gpurun_out/r05q_k4ab.log
GF_K4_QB=2 GF_K4_WV=4 python -m pytest tests/test_ops_gpu.py -k self_attention -q -m gpu 2>&1 | tail -2 >> gpurun_out/r05q_k4ab.log
cat gpurun_out/r05q_k4ab.log
python -m pytest tests/test_e2e_gpu.py tests/test_outcome_parity_gpu.py -m gpu -q -s > gpurun_out/r05q_pytest.log 2>&1
tail -4 gpurun_out/r05q_pytest.log
