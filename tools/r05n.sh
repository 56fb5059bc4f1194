python -m pytest tests/test_train_gpu.py tests/test_train_hip_backward.py -m gpu -q -x > gpurun_out/r05n_train.log 2>&1
python tools/train_profile.py 640 640 --bf16 --hip --batch 8 --mega --no-prof > gpurun_out/r05n_mega.log 2>&1
python tools/train_profile.py 640 640 --bf16 --hip --batch 2 --long --no-prof > gpurun_out/r05n_b2.log 2>&1
python tools/k4_ab.py 1195 1,8 1,8,msum > gpurun_out/r05n_k4.log 2>&1
tail -3 gpurun_out/r05n_train.log; grep ^step gpurun_out/r05n_mega.log | tail -2; grep ^step gpurun_out/r05n_b2.log | tail -2; cat gpurun_out/r05n_k4.log
