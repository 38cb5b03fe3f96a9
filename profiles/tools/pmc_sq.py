"""Scratch: MFMA-busy fraction and clock under load per kernel from a rocprofv3 --pmc pass
(GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY).
    python profiles/tools/pmc_sq.py counter_collection.csv kernel_trace_or_'-' out.json substr [substr ...]"""
import collections, csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    if 'Start_Timestamp' in r and r.get('End_Timestamp'):
        dur[(r['Kernel_Name'], r['Dispatch_Id'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
out = {}
for k, d in agg.items():
    if not any(s in k for s in sys.argv[4:]):
        continue
    c = {n: sum(v) / len(v) for n, v in d.items()}
    ns = [v for (kk, _), v in dur.items() if kk == k]
    avg_ns = sum(ns) / len(ns) if ns else None
    name = k.split('(')[0].replace('void ', '')
    rec = {'launches': len(next(iter(d.values()))), 'avg_ns_under_pmc': avg_ns, 'counters': c}
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        rec['mfma_busy_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)      # 8 XCDs summed; 1024 SIMDs
        if avg_ns:
            rec['clock_GHz'] = c['GRBM_GUI_ACTIVE'] / 8 / avg_ns
    out[name] = rec
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
