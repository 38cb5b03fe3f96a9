"""Scratch: turn the round-3 rocprofv3 outputs under gpurun_out/ into the summaries committed under profiles/.
    python profiles/tools/r03_collect.py"""
import collections, csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)      # (older runs stay in gpurun_out/)
    assert f, pattern
    return f[-1]


def short(name):
    name = name.replace('void ', '').replace('hnrf::', '')
    return name.split('(')[0][:70]


def pmc(pattern, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(pattern))):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


shutil.copy(one('r03_main/*/*_kernel_stats.csv'), os.path.join(P, 'r03_f16x3_kernel_stats.csv'))
shutil.copy(one('r03_main32/*/*_kernel_stats.csv'), os.path.join(P, 'r03_f32_kernel_stats.csv'))
shutil.copy(one('r03_train/*/*_kernel_stats.csv'), os.path.join(P, 'r03_train_kernel_stats.csv'))
for src, dst in (('r03_bench_f16x3.json', 'r03_bench_f16x3.json'), ('r03_bench_f32.json', 'r03_bench_f32.json'),
                 ('r03_bench_default.json', 'r03_bench_default.json')):
    line = [l for l in open(os.path.join(G, src)).read().splitlines() if l.startswith('{')][-1]
    open(os.path.join(P, dst), 'w').write(line + '\n')

# headline kernel: HBM bytes per launch (guide's recipe: FETCH_SIZE x2 on gfx950 for wide coalesced reads, WRITE_SIZE as is; KB units)
f = pmc('r03_pmc_fetch/*/*_counter_collection.csv', 'FETCH_SIZE')
w = pmc('r03_pmc_write/*/*_counter_collection.csv', 'WRITE_SIZE')
# (two instances of the inference kernel run: <0, true> = with the f16-range guard, <0, false> = without; same traffic)
ks = [n for n in f if 'canonical_f16x3_kernel<0' in n]
fv, wv = [v for n in ks for v in f[n]], [v for n in ks for v in w[n]]
fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
f = {ks[0]: fv}
k = ks[0]
json.dump({
    'kernel': 'canonical_f16x3_kernel', 'samples_per_launch': 4194304, 'launches_averaged': len(f[k]),
    'FETCH_SIZE_KB_raw': fk, 'WRITE_SIZE_KB_raw': wk,
    'correction': 'FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B for wide coalesced reads, MI355X_MICROARCH.md '
                  'HBM section); WRITE_SIZE as is',
    'hbm_bytes_per_launch': (2 * fk + wk) * 1024, 'algorithmic_bytes_per_launch': 4194304 * 28,
    'note': 'separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of: rocprofv3 --pmc X -- python3 bench.py --steps 2 --warmup 1 --main-only',
}, open(os.path.join(P, 'r03_pmc_canonical.json'), 'w'), indent=1)

# training step: per-kernel time (kernel stats of 6 iterations) and HBM bytes (PMC passes of 3 iterations)
tf = pmc('r03_pmc_tfetch/*/*_counter_collection.csv', 'FETCH_SIZE')
tw = pmc('r03_pmc_twrite/*/*_counter_collection.csv', 'WRITE_SIZE')
stats = {r['Name']: r for r in csv.DictReader(open(one('r03_train/*/*_kernel_stats.csv')))}
ITERS_STATS, ITERS_PMC = 16, 13                       # time_train.py: N printed iterations + 10 timed ones
rows = []
for name, r in stats.items():
    calls, tot = int(r['Calls']), float(r['TotalDurationNs'])
    fr = sum(tf.get(name, [0.])) * 2 * 1024 / ITERS_PMC
    wr = sum(tw.get(name, [0.])) * 1024 / ITERS_PMC
    rows.append({'kernel': short(name), 'calls_per_iter': round(calls / ITERS_STATS, 2), 'ms_per_iter': round(tot / ITERS_STATS / 1e6, 4),
                 'hbm_read_MB_per_iter': round(fr / 1e6, 1), 'hbm_write_MB_per_iter': round(wr / 1e6, 1),
                 'TB_per_s': round((fr + wr) / (tot / ITERS_STATS * 1e-9) / 1e12, 2) if tot else None})
rows.sort(key=lambda r: -r['ms_per_iter'])
tot_ms = sum(r['ms_per_iter'] for r in rows)
tot_rd, tot_wr = sum(r['hbm_read_MB_per_iter'] for r in rows), sum(r['hbm_write_MB_per_iter'] for r in rows)
with open(os.path.join(P, 'r03_train_traffic.csv'), 'w') as fo:
    wtr = csv.DictWriter(fo, fieldnames=list(rows[0]))
    wtr.writeheader()
    for r in rows[:40]:
        wtr.writerow(r)
    wtr.writerow({'kernel': 'TOTAL (all %d kernels)' % len(rows), 'calls_per_iter': '', 'ms_per_iter': round(tot_ms, 3),
                  'hbm_read_MB_per_iter': round(tot_rd, 1), 'hbm_write_MB_per_iter': round(tot_wr, 1),
                  'TB_per_s': round((tot_rd + tot_wr) * 1e6 / (tot_ms * 1e-3) / 1e12, 2)})
for r in rows[:24]:
    print(r)
print('total ms %.2f  read %.0f MB  write %.0f MB' % (tot_ms, tot_rd, tot_wr))


# rendering frame: per kernel average launch time, HBM bytes per launch, MFMA busy / clock (separate PMC passes of the same command)
import subprocess, sys
sq_json = os.path.join(P, 'r03_pmc_sq_f16x3_summary.json')
subprocess.run([sys.executable, os.path.join(P, 'tools', 'pmc_sq.py'), one('r03_pmc_sq/*/*_counter_collection.csv'), '-', sq_json,
                'canonical_f16x3', 'nonrigid_f16x3', 'sample_warp', 'composite'], check=True, stdout=subprocess.DEVNULL)
tsq = glob.glob(os.path.join(G, 'r03_pmc_tsq/*/*_counter_collection.csv'))
if tsq:
    subprocess.run([sys.executable, os.path.join(P, 'tools', 'pmc_sq.py'), one('r03_pmc_tsq/*/*_counter_collection.csv'), '-',
                    os.path.join(P, 'r03_pmc_sq_train_summary.json'), 'canonical', 'nonrigid', 'mlp_dwh', 'sample_warp'],
                   check=True, stdout=subprocess.DEVNULL)
sq = json.load(open(sq_json))
f = pmc('r03_pmc_fetch/*/*_counter_collection.csv', 'FETCH_SIZE')
w = pmc('r03_pmc_write/*/*_counter_collection.csv', 'WRITE_SIZE')
fstats = {r['Name']: r for r in csv.DictReader(open(one('r03_main/*/*_kernel_stats.csv')))}
frows = []
for name, r in fstats.items():
    if not any(k in name for k in ('canonical_f16x3', 'nonrigid_f16x3', 'sample_warp', 'composite_kernel')):
        continue
    fr = sum(f.get(name, [0.])) / max(len(f.get(name, [0.])), 1) * 2 * 1024
    wr = sum(w.get(name, [0.])) / max(len(w.get(name, [0.])), 1) * 1024
    key = name.split('(')[0].replace('void ', '')
    rec = sq.get(key, {})
    ms = float(r['AverageNs']) / 1e6
    frows.append({'kernel': short(name), 'launches': r['Calls'], 'avg_ms': round(ms, 4), 'hbm_read_MB': round(fr / 1e6, 1),
                  'hbm_write_MB': round(wr / 1e6, 1), 'TB_per_s': round((fr + wr) / (ms * 1e-3) / 1e12, 3),
                  'mfma_busy': round(rec.get('mfma_busy_frac', 0.0), 3), 'clock_GHz': round(rec.get('clock_GHz') or 0.0, 2)})
frows.sort(key=lambda r: -r['avg_ms'])
with open(os.path.join(P, 'r03_frame_traffic.csv'), 'w') as fo:
    wtr = csv.DictWriter(fo, fieldnames=list(frows[0]))
    wtr.writeheader()
    for r in frows:
        wtr.writerow(r)
for r in frows:
    print(r)
