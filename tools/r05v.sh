( time python -m pytest tests/test_e2e_gpu.py tests/test_ops_gpu.py tests/test_stress_gpu.py tests/test_outcome_parity_gpu.py tests/test_matcher_gpu.py -m gpu -q ) > gpurun_out/r05v_pytest.log 2>&1
python bench.py > gpurun_out/r05v_bench.json 2> gpurun_out/r05v_bench.err
bash tools/gpu_profile.sh r05v > gpurun_out/r05v_profile.log 2>&1
tail -5 gpurun_out/r05v_pytest.log
tail -2 gpurun_out/r05v_bench.err
