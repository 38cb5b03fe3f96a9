"""Scratch: per-kernel averages of rocprofv3 --pmc counter CSVs (python profiles/tools/pmc_summary.py file.csv substring ...)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print(k[:100], {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
