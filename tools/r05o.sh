python tools/train_profile.py 640 640 --bf16 --hip --batch 8 --mega > gpurun_out/r05o_mega.log 2>&1
echo "== default build" > gpurun_out/r05o_prio.log
python tools/fine_layer_time.py >> gpurun_out/r05o_prio.log 2>&1
python tools/k4_ab.py 1195 1,8 >> gpurun_out/r05o_prio.log 2>&1
HIPCC_EXTRA="-DK11_PRIO -DK4_PRIO" python -m geoformer_amd.build -q >> gpurun_out/r05o_prio.log 2>&1
echo "== -DK11_PRIO -DK4_PRIO" >> gpurun_out/r05o_prio.log
python tools/fine_layer_time.py >> gpurun_out/r05o_prio.log 2>&1
python tools/k4_ab.py 1195 1,8 >> gpurun_out/r05o_prio.log 2>&1
grep -v amdgpu.ids gpurun_out/r05o_prio.log
grep -E "^step" gpurun_out/r05o_mega.log | tail -2
