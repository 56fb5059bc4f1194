#!/bin/bash
# kernel trace of the bf16 bench configuration (single stream), to compare kernel by kernel with the fp16 trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06bf_trace -o run -- python3 bench.py --streams 1 --no-cpu-baseline --no-extras --precision bf16 --steps 10 --warmup 3 > $OUT/r06bf_trace.json 2> $OUT/r06bf_trace.err
find $OUT/r06bf_trace -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
python3 tools/prof_summary.py $OUT/r06bf_trace > $OUT/r06bf_trace_summary.txt 2>&1
head -45 $OUT/r06bf_trace_summary.txt | cut -c1-150
