# the round's closing run: whole GPU suite, default bench line, bf16 bench line, profile set
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05p_pytest.log 2>&1
python bench.py > gpurun_out/r05p_bench.json 2> gpurun_out/r05p_bench.err
python bench.py --precision bf16 --no-train --no-cpu-baseline > gpurun_out/r05p_bench_bf16.json 2> gpurun_out/r05p_bench_bf16.err
bash tools/gpu_profile.sh r05p > gpurun_out/r05p_profile.log 2>&1
tail -6 gpurun_out/r05p_pytest.log
tail -2 gpurun_out/r05p_bench.err
tail -3 gpurun_out/r05p_profile.log | cut -c1-300
