#!/bin/bash
# SQ counters of the K9 kernels alone (tools/k9_time.py 16): bash tools/k9_pmc.sh  (through gpurun, from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/k9_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/a -o run -- python3 tools/k9_time.py 16 3 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d $OUT/b -o run -- python3 tools/k9_time.py 16 3 > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ('a', 'b'):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob('$OUT/' + tag + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            key = 'enc_layer<1,true>' if 'enc_layer' in n and 'Lb1' in n else 'enc_layer<finish>' if 'enc_layer' in n else 'enc_kv_state' if 'enc_kv_state' in n else None
            if key:
                a = acc[key][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, d in acc.items():
        print(k, {c: round(v[0] / max(v[1], 1)) for c, v in d.items()})
PY
