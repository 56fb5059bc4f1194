#!/bin/bash
# rocprofv3 kernel trace of a few MegaDepth-style training steps (batch 8, bf16, HIP Functions + HipConv3x3): per-kernel totals
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06tr_trace -o run -- python3 tools/train_profile.py 640 640 --bf16 --hip --batch 8 --mega --hipconv --no-prof --long > $OUT/r06tr_trace.log 2> $OUT/r06tr_trace.err
find $OUT/r06tr_trace -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
python3 tools/prof_summary.py $OUT/r06tr_trace > $OUT/r06tr_trace_summary.txt 2>&1
grep "^step" $OUT/r06tr_trace.log | tail -4
head -48 $OUT/r06tr_trace_summary.txt | cut -c1-150
