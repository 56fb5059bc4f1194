#!/bin/bash
# SQ counters per kernel of a small driver script:  bash tools/kernel_pmc.sh <name filter (substring)> <python script> [args...]
# (through gpurun, from the repo root; the program follows `--` directly, see tools/gpu_profile.sh)
FILTER=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kernel_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/a -o run -- python3 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d $OUT/b -o run -- python3 "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM \
    --output-format csv -d $OUT/c -o run -- python3 "$@" > $OUT/c.log 2>&1
FILTER="$FILTER" OUT="$OUT" python3 - <<'PY'
import csv, glob, collections, os
flt, out = os.environ['FILTER'], os.environ['OUT']
for tag in ('a', 'b', 'c'):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(out + '/' + tag + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if flt in n:
                a = acc[n[:60]][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, d in acc.items():
        print(k, {c: round(v[0] / max(v[1], 1)) for c, v in d.items()})
PY
