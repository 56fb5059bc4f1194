python tools/k4_ab.py 1195 1,8 1,8,vsum 1,8 1,8,vsum > gpurun_out/r05m_k4ab.log 2>&1
python -m pytest tests/test_encoder_fused.py tests/test_fine_layer.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/r05m_k4ab.log
python tools/k9_time.py 16 >> gpurun_out/r05m_k4ab.log 2>&1
python tools/k9_time.py 8 >> gpurun_out/r05m_k4ab.log 2>&1
python tools/fine_layer_time.py >> gpurun_out/r05m_k4ab.log 2>&1
python tools/train_profile.py 640 640 --bf16 --hip --batch 8 --mega > gpurun_out/r05m_train_mega.log 2>&1
cat gpurun_out/r05m_k4ab.log
grep -E "^step|Self CUDA time|device time" gpurun_out/r05m_train_mega.log | tail -8
