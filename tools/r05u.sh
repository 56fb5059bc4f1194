python tools/k4_ab.py 1195 1,8 1,8,gather 1,8 1,8,gather > gpurun_out/r05u_k4.log 2>&1
python tools/k4_ab.py 333 1,8 1,8,gather >> gpurun_out/r05u_k4.log 2>&1
python tools/k4_ab.py 70 1,8 1,8,gather >> gpurun_out/r05u_k4.log 2>&1
python -m pytest tests/test_ops_gpu.py -k "self_attention" -q -m gpu -s 2>&1 | tail -6 >> gpurun_out/r05u_k4.log
python __graft_entry__.py smoke 2>&1 | tail -2 >> gpurun_out/r05u_k4.log
cat gpurun_out/r05u_k4.log
