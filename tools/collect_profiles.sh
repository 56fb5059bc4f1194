#!/bin/bash
# Copies the summaries of a tools/gpu_profile.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
#     bash tools/collect_profiles.sh <gpurun tag> <profiles prefix>       e.g.  r02c r02 ;  r02c_nominal r02_nominal
set -eu
SRC=gpurun_out; TAG=$1; DST=profiles/$2
cp $SRC/${TAG}_trace_summary.txt ${DST}_kernel_stats.txt
cp $SRC/${TAG}_trace/run_kernel_stats.csv ${DST}_kernel_stats.csv
cp $SRC/${TAG}_pmc_summary.txt ${DST}_pmc_summary.txt
cp $SRC/${TAG}_pmc_per_tag.json ${DST}_pmc_per_tag.json
tail -1 $SRC/${TAG}_trace.json > ${DST}_trace_bench_line.json
ls -la ${DST}_*
