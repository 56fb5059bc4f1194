#!/bin/bash
# SQ counters of the K10 kernels at the shapes of tools/k10_time.py:  bash tools/k10_pmc.sh   (through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/k10_pmc; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp; cd $ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT -o run -- python3 tools/k10_time.py > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/k10_pmc/**/*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv3x3' in r['Kernel_Name']:
            k = r['Kernel_Name'][30:62]
            a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, c in acc.items():
    g = {n: v[0] / max(v[1], 1) for n, v in c.items()}
    wc = g.get('SQ_WAVE_CYCLES', 1)
    print(k, 'calls', int(c['SQ_WAVE_CYCLES'][1]),
          ' wait_any %.2f wait_inst %.2f active %.2f' % (g['SQ_WAIT_ANY'] / wc, g['SQ_WAIT_INST_ANY'] / wc, g['SQ_ACTIVE_INST_ANY'] / wc),
          ' lds_conflict/lds_active %.3f' % (g['SQ_LDS_BANK_CONFLICT'] / max(g['SQ_LDS_IDX_ACTIVE'], 1)),
          ' lds_active/gui %.3f' % (g['SQ_LDS_IDX_ACTIVE'] / (g['GRBM_GUI_ACTIVE'] / 8 * 256)),
          ' mfma_util %.3f' % (g['SQ_VALU_MFMA_BUSY_CYCLES'] / (g['GRBM_GUI_ACTIVE'] / 8 * 1024)))
PY
rm -rf $OUT
