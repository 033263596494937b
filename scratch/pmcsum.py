"""Per-kernel means of the counters in rocprofv3 --pmc output directories: python scratch/pmcsum.py <substr> <dir>..."""
import csv, glob, sys, collections
sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sub in r['Kernel_Name']:
                acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            print(f'{k:60s} {c:28s} {sum(v)/len(v):16.1f}  n={len(v)}')
