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


_DEMANGLED = {}


def family(name):
    """Kernel family = the function name without namespace, return type and argument list; template arguments kept
    (`linear_kernel<_Float16, 8, 4>`); mangled names (rocprofv3 leaves some) go through c++filt."""
    if name.startswith('_Z'):
        if name not in _DEMANGLED:
            import subprocess
            try:
                _DEMANGLED[name] = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                _DEMANGLED[name] = name
        name = _DEMANGLED[name]
        m = re.match(r'^_ZN12_GLOBAL__N_1(\d+)', name)
        if m:      # c++filt does not know _Float16 (DF16_): decode `name<template args>` of our own kernels by hand
            n = int(m.group(1))
            ident, rest = name[m.end():m.end() + n], name[m.end() + n:]
            args = []
            if rest.startswith('I'):
                for tok in re.findall(r'DF16_|DF16b|Li\d+E|Lb\dE|f', rest[1:rest.index('EE') + 1] if 'EE' in rest else rest[1:]):
                    args.append({'DF16_': '_Float16', 'DF16b': '__bf16', 'f': 'float'}.get(tok) or
                                (tok[2:-1] if tok.startswith('Li') else ('true' if tok == 'Lb1E' else 'false')))
            name = ident + ('<' + ', '.join(args) + '>' if args else '')
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    depth, out = 0, []
    for ch in name:                      # cut at the '(' that opens the argument list (template args may hold parentheses)
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return ''.join(out)[:72]


# bench.py's roofline tags -> the kernel families a tagged call launches (substring match on the family name)
TAG_FAMILIES = {
    'enc_layer': ['enc_layer<'],
    'enc_kv_state': ['enc_kv_state<', 'enc_kv_reduce'],
    'k3_linear': ['linear_kernel'],              # minus TAG_EXCLUDE: the EPI_UPADD instance belongs to the backbone
    'k3_upadd': [', 4, 5, true>'],                # linear_kernel<T, 4, EPI_UPADD = 5, CONVX>
    'conv1x1': [', 4, 0, true>'],                 # linear_kernel<T, 4, EPI_NONE, CONVX>: gf_conv1x1_nhwc
    'k1_stats': ['k1_stats'],
    'k1_conf': ['k1_conf'],
    'k1_unit': ['k1_'],
    'fine_layer': ['fine_layer<', 'fine_kv'],
    'k4_self_attention': ['attn_self<', 'attn_self_head<', 'attn_gather_kv'],
    'k2_linear_attention': ['la16_', 'la_kv_', 'la_apply', 'la_small'],
    'k5_window_attention': ['window_cross_attention', 'window_cross_tiled'],
    'bias_act': ['bias_act<'],
    'conv3x3': ['conv3x3_kernel'],
}
TAG_EXCLUDE = {'k3_linear': [', 4, 5, true>', ', 4, 0, true>']}
# the family whose call count equals the number of tagged calls
TAG_PRIMARY = {'k1_unit': 'k1_compact', 'fine_layer': 'fine_layer<', 'k4_self_attention': ('attn_self_head<', 'attn_self<'), 'enc_layer': 'enc_layer<', 'enc_kv_state': 'enc_kv_state<', 'k3_linear': 'linear_kernel', 'k3_upadd': ', 4, 5, true>', 'conv1x1': ', 4, 0, true>', 'k1_stats': 'k1_stats',
               'k1_conf': 'k1_conf', 'k2_linear_attention': ('la16_apply', 'la_apply', 'la_small'), 'k5_window_attention': ('window_cross_tiled', 'window_cross_attention'),
               'bias_act': 'bias_act<', 'conv3x3': 'conv3x3_kernel'}


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
    # per bench.py tag: PMC HBM bytes per tagged call and MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES summed over the
    # 1024 SIMDs / (GRBM_GUI_ACTIVE per XCD x 1024): the gfx9 MfmaUtil expression of rocprofv3's counter definitions)
    per_tag = {}
    print()
    print(f'{"tag":22s} {"calls":>6s} {"avg_us":>8s} {"HBM MB/call":>12s} {"GB/s":>7s} {"mfma_util":>9s}')
    for btag, pats in TAG_FAMILIES.items():
        fam = [r for r in rows if any(p in r['kernel'] for p in pats) and not any(x in r['kernel'] for x in TAG_EXCLUDE.get(btag, []))]
        if not fam:
            continue
        prim = TAG_PRIMARY[btag]
        prim = prim if isinstance(prim, tuple) else (prim,)
        calls = sum(r['calls'] for r in fam if any(p in r['kernel'] for p in prim)) or 1
        t_us = sum(r['calls'] * r['avg_us'] for r in fam)
        hbm = sum(r['calls'] * r['hbm_bytes'] for r in fam if 'hbm_bytes' in r)
        mb = sum(r['calls'] * r.get('mfma_busy_cycles', 0.0) for r in fam)
        ga = sum(r['calls'] * r.get('gui_active', 0.0) for r in fam)
        util = mb / (ga / 8.0 * 1024.0) if ga else None
        per_tag[btag] = {'calls': calls, 'avg_us_per_call': t_us / calls, 'hbm_bytes_per_launch': hbm / calls if hbm else None,
                        'mfma_util': util, 'families': sorted({r['kernel'] for r in fam})}
        print(f'{btag:22s} {calls:6d} {t_us / calls:8.1f} {hbm / calls / 1e6:12.1f} {hbm / max(t_us, 1e-9) * 1e-3:7.0f} '
              f'{(util if util is not None else float("nan")):9.3f}')
    if '--json' in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)
    if '--tags' in sys.argv:
        # the configuration these counters were recorded for = the bench line the traced run printed (bench.py attaches
        # the figures only to runs of the same configuration and names this file as their source)
        try:
            import datetime
            line = [ln for ln in open(os.path.join(root, tag + '_trace.json')).read().splitlines() if ln.startswith('{')][-1]
            cfg = json.loads(line)['config']
            per_tag['_config'] = {k: cfg.get(k) for k in ('precision', 'batch', 'size', 'pairs', 'coarse_thr', 'fine_thr')}
            per_tag['_recorded'] = datetime.date.today().isoformat()
        except Exception as e:                      # noqa: BLE001
            per_tag['_config'] = None
            print('no bench line for the _config block:', e)
        json.dump(per_tag, open(sys.argv[sys.argv.index('--tags') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
