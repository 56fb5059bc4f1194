#!/bin/bash
# Profiles of one bench.py configuration on the GPU box (run through gpurun from the repo root):
#     bash tools/gpu_profile.sh <tag> [extra bench.py args]
# writes gpurun_out/<tag>_{trace,pmc_sq,pmc_fetch,pmc_write}/ and the summaries tools/prof_summary.py /
# tools/pmc_roofline.py read.  The kernel trace is a SINGLE-STREAM run (--streams 1), so the average
# launch durations it reports are the ones the bench line's roofline entries quote.
# rocprofv3 gets `python3 bench.py ...` directly after `--` (no env/bash/shebang hop: the profiler's
# preloaded library initialises the GPU before the program starts).
set -u
TAG=${1:-r02}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
BENCH="python3 bench.py --streams 1 --no-cpu-baseline --no-extras $*"
echo "== kernel trace: $BENCH"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- $BENCH --steps 10 --warmup 3 > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.err
echo "rc=$?"
SHORT="$BENCH --steps 2 --warmup 1"
echo "== pmc SQ"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/${TAG}_pmc_sq -o run -- $SHORT > $OUT/${TAG}_pmc_sq.json 2> $OUT/${TAG}_pmc_sq.err
echo "rc=$?"
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o run -- $SHORT > $OUT/${TAG}_pmc_fetch.json 2> $OUT/${TAG}_pmc_fetch.err
echo "rc=$?"
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o run -- $SHORT > $OUT/${TAG}_pmc_write.json 2> $OUT/${TAG}_pmc_write.err
echo "rc=$?"
# keep the merged-back payload small: only the per-kernel csv files are needed
find $OUT/${TAG}_trace $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write -type f ! -name '*.csv' -delete 2>/dev/null
python3 tools/prof_summary.py $OUT/${TAG}_trace > $OUT/${TAG}_trace_summary.txt 2>&1
python3 tools/pmc_roofline.py $OUT $TAG --tags $OUT/${TAG}_pmc_per_tag.json > $OUT/${TAG}_pmc_summary.txt 2>&1
# the raw per-dispatch counter csvs are ~30 MB per pass (gpurun merges back at most 64 MiB): keep them only on request
if [ -z "${KEEP_RAW:-}" ]; then
    rm -rf $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
    find $OUT/${TAG}_trace -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
fi
tail -5 $OUT/${TAG}_trace.err
cat $OUT/${TAG}_trace.json | tail -1 | cut -c1-600
