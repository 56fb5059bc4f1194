"""Compact view of a rocprofv3 kernel_stats.csv: python tools/prof_summary.py <dir-or-csv> [topN]"""
import csv, glob, os, re, sys
p = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
files = [p] if p.endswith('.csv') else glob.glob(os.path.join(p, '**', '*kernel_stats.csv'), recursive=True)
for f in files:
    rows = list(csv.DictReader(open(f)))
    print(f'== {f}')
    print(f'{"kernel":60s} {"calls":>6s} {"avg_us":>10s} {"total_ms":>10s} {"%":>6s}')
    for r in rows[:top]:
        name = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
        name = re.sub(r'^void ', '', name)[:60]
        print(f'{name:60s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:10.2f} {float(r["TotalDurationNs"]) / 1e6:10.3f} {float(r["Percentage"]):6.2f}')
