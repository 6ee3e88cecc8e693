#!/bin/bash
# kernel-time A/B of k_tri_matvec_rb variants (one-pass predict-with-gradients): bash scripts/grad_rb_sweep.sh [N] [d] "opts" "opts" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-8192}; D=${2:-8}; shift; shift
for o in "$@"; do
  rm -rf /tmp/gt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gt -o g -- python $R/scripts/grad_call_probe.py $N $D $o > /tmp/gt.log 2>&1
  f=$(find /tmp/gt -name "*kernel_stats.csv" | head -1)
  echo "== $o"; grep "rows= 1" /tmp/gt.log
  python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_tri_matvec', 'k_part_sum', 'k_grad_reduce', 'k_kstar', 'copyBuffer')):
        print('   %-40s calls %5s avg %8.1f us' % (n.split('(')[0][-40:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
