#!/bin/bash
# round 6, last pass: the whole GPU suite, smoke, the default and bf16 bench lines on the final code (kernel trace / PMC: tools/r06_final.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest.log 2>&1; tail -4 gpurun_out/r06_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -14 gpurun_out/r06_bench_default.err
timeout 600 python bench.py --precision bf16 --no-train --no-cpu-baseline > gpurun_out/r06_bench_bf16.json 2> gpurun_out/r06_bench_bf16.err; tail -4 gpurun_out/r06_bench_bf16.err
