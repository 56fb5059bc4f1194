echo "== default" > gpurun_out/r05t_k10.log
python tools/k10_time.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05t_k10.log
for n in 1 2 4; do
HIPCC_EXTRA="-DK10_STAGGER=$n" python -m geoformer_amd.build -q > /dev/null 2>&1
echo "== -DK10_STAGGER=$n" >> gpurun_out/r05t_k10.log
python tools/k10_time.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05t_k10.log
done
cat gpurun_out/r05t_k10.log
