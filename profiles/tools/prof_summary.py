"""Scratch: per-step kernel summary from a rocprofv3 rocpd database (python profiles/tools/prof_summary.py db [steps])."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
suf = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = db.execute(f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id=s.id order by d.start").fetchall()
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if 'sample_warp_kernel' in n and 'bwd' not in n]
lo, hi = idx[-nsteps - 1], idx[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows[lo:hi]:
    agg[n][0] += 1
    agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
print('steps %d  GPU busy per step %.3f ms  kernels per step %.0f' % (nsteps, tot / nsteps / 1e6, sum(v[0] for v in agg.values()) / nsteps))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print('%-84s x%-5.1f %8.3f ms/step' % (n[:84], c / nsteps, t / nsteps / 1e6))
