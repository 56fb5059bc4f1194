"""Average PMC counter values per kernel from rocprofv3 --pmc output:
   python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if pat in k:
            key = (k[:70], r['Counter_Name'])
            acc[key][0] += float(r['Counter_Value']); acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f'{k:70s} {c:14s} avg {s / n:16.1f} over {n} dispatches')
