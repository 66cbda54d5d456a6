"""Per-kernel means of rocprofv3 counter CSVs (one row per kernel instantiation and grid size)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for p in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p, newline='')):
            k = r['Kernel_Name']
            if sys.argv[1] not in k:
                continue
            key = (k.split('(')[0].replace('void ', ''), r['Grid_Size'])
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k)
    for c, x in sorted(v.items()):
        print('    %-28s %s' % (c, ' '.join('%10.2f' % (t / 1e6) for t in x[:4])))
