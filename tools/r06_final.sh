#!/bin/bash
# round 6 final evidence: the whole GPU suite, smoke, the default bench line, the bf16 line, the rocprofv3 kernel trace + PMC passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest.log 2>&1; tail -4 gpurun_out/r06_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -14 gpurun_out/r06_bench_default.err
timeout 600 python bench.py --precision bf16 --no-train --no-cpu-baseline > gpurun_out/r06_bench_bf16.json 2> gpurun_out/r06_bench_bf16.err; tail -3 gpurun_out/r06_bench_bf16.err
bash tools/gpu_profile.sh r06 > gpurun_out/r06_profile.log 2>&1; tail -3 gpurun_out/r06_profile.log | cut -c1-300
