import csv, glob, sys
f = sorted(glob.glob(sys.argv[1]))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print('%-80s %5s %10.4f ms avg  %6s%%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e6, r['Percentage'][:5]))
