#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) and SQ counters of one K4 form:  bash tools/k4_prof.sh <tag>   (GF_K4_* in the environment)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/k4_prof_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o run -- python3 tools/k4_time.py 1195 > $OUT/t.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attn' in r['Name']:
            print('$1', r['Name'][:70], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1))
PY
bash tools/kernel_pmc.sh attn_self tools/k4_time.py 1195
