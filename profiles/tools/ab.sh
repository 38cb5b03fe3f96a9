#!/bin/bash
# usage: ab.sh variant1 variant2 ...   (libraries profiles/tools/libhnrf_ab_<variant>.so)
for i in 1 2; do for v in "$@"; do
  cd /tmp; export TMPDIR=/tmp
  rm -rf /tmp/abp_$v; HNRF_LIB_PATH=$GRAFT_REPO_ROOT/profiles/tools/libhnrf_ab_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -o m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --main-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], end=' ')"
  python3 - <<PY
import csv
rows={r['Name'][:44]:float(r['AverageNs'])/1e6 for r in csv.DictReader(open('/tmp/abp_$v/m_kernel_stats.csv'))}
print(' '.join('%s=%.4f' % (k.replace('void hnrf::','')[:28], v) for k,v in rows.items() if ('x2_kernel<false' in k or 'canonical_f16x3_kernel<0, false' in k or 'sample_warp' in k)))
PY
done; done
