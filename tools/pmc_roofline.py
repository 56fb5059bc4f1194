"""Per-kernel roofline evidence from the rocprofv3 passes tools/gpu_profile.sh writes:

    python tools/pmc_roofline.py <gpurun_out dir> <tag> [--json out.json]

For every kernel family (name prefix before the first '(' / template list, anonymous namespace stripped):
  * calls and average duration from the single-stream kernel trace,
  * MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (both summed over the dispatches of the family;
    SQ_BUSY_CYCLES counts per-SE busy cycles, SQ_VALU_MFMA_BUSY_CYCLES the cycles a SIMD's matrix pipe is busy:
    the quotient is reported as measured, plus the per-dispatch MFMA instruction count),
  * HBM traffic per dispatch: FETCH_SIZE x 2 (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md
    'HBM') + WRITE_SIZE, both reported by rocprofv3 in KiB.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def family(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0][:64]


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[family(r['Kernel_Name'])][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    return acc


def trace(d):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = family(r['Name'])
            c, t = out.get(k, (0, 0.0))
            out[k] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
    return out


def main():
    root, tag = sys.argv[1], sys.argv[2]
    tr = trace(os.path.join(root, tag + '_trace'))
    sq = counters(os.path.join(root, tag + '_pmc_sq'))
    fe = counters(os.path.join(root, tag + '_pmc_fetch'))
    wr = counters(os.path.join(root, tag + '_pmc_write'))
    total = sum(t for _, t in tr.values()) or 1.0
    rows = []
    for k, (calls, tns) in sorted(tr.items(), key=lambda kv: -kv[1][1]):
        row = {'kernel': k, 'calls': calls, 'avg_us': tns / calls / 1e3, 'share': tns / total}
        s = sq.get(k)
        if s:
            mb, bc = s['SQ_VALU_MFMA_BUSY_CYCLES'][0], s['SQ_BUSY_CYCLES'][0]
            n = max(s['SQ_BUSY_CYCLES'][1], 1)
            row.update(mfma_busy_cycles=mb / n, sq_busy_cycles=bc / n, mfma_insts=s['SQ_INSTS_MFMA'][0] / n,
                       wave_cycles=s['SQ_WAVE_CYCLES'][0] / n, wait_any=s['SQ_WAIT_ANY'][0] / n,
                       wait_inst_any=s['SQ_WAIT_INST_ANY'][0] / n, active_inst_any=s['SQ_ACTIVE_INST_ANY'][0] / n,
                       gui_active=s['GRBM_GUI_ACTIVE'][0] / n)
        f, w = fe.get(k), wr.get(k)
        if f and w:
            fb = f['FETCH_SIZE'][0] / max(f['FETCH_SIZE'][1], 1) * 1024.0
            wb = w['WRITE_SIZE'][0] / max(w['WRITE_SIZE'][1], 1) * 1024.0
            row.update(fetch_bytes_raw=fb, write_bytes=wb, hbm_bytes=2.0 * fb + wb)
        rows.append(row)
    print(f'{"kernel":44s} {"calls":>5s} {"avg_us":>8s} {"share":>6s} {"mfma_insts":>11s} {"mfma_busy/sq_busy":>17s} {"HBM MB":>9s} {"GB/s":>7s}')
    for r in rows[:40]:
        mu = r['mfma_busy_cycles'] / r['sq_busy_cycles'] if r.get('sq_busy_cycles') else float('nan')
        hb = r.get('hbm_bytes', float('nan'))
        print(f'{r["kernel"][:44]:44s} {r["calls"]:5d} {r["avg_us"]:8.1f} {r["share"] * 100:5.1f}% {r.get("mfma_insts", float("nan")):11.0f} '
              f'{mu:17.3f} {hb / 1e6:9.1f} {hb / (r["avg_us"] * 1e-6) / 1e9:7.0f}')
    if '--json' in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
