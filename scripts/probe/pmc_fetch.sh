#!/bin/bash
# L2 -> fabric read bytes (rocprofv3 --pmc FETCH_SIZE, doubled per MI355X_MICROARCH.md) of the sweep schedules in the stand-alone
# A/B binary: bash scripts/probe/pmc_fetch.sh N cols "tile_order[:super_m] ..."      (one launch + two warm-ups per variant)
cd "$(dirname "$0")"
N=${1:-8192}; COLS=${2:-65536}; shift 2
O=../../gpurun_out/r06/pmc_fetch; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OLDPWD/$O -o p -- $OLDPWD/sweep_ab.bin $N $COLS 1 "$@" > $OLDPWD/$O/run.log 2>&1
cd $OLDPWD
f=$(find $O -name "*counter_collection.csv" | head -1)
grep RESULT $O/run.log | cut -c1-120
python3 - "$f" "$#" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Counter_Name'] == 'FETCH_SIZE' and 'sweep' in r['Kernel_Name']]
nvar = int(sys.argv[2])
vals = [float(r['Counter_Value']) for r in rows]
names = [r['Kernel_Name'][:60] for r in rows]
if len(vals) == 3 * nvar:            # three launches per variant, in order
    for i in range(0, len(vals), 3):
        v = vals[i:i + 3]
        print('%-62s FETCH_SIZE x2 = %.1f GB per launch (raw KiB %s)' % (names[i], 2 * 1024 * v[-1] / 1e9, [int(x) for x in v]))
else:                                # a variant that splits its launch: all of the run's dispatches together
    print('%-62s FETCH_SIZE x2 = %.1f GB per sweep call (%d dispatches, %d variant(s) x 3 calls)' % (names[0], 2 * 1024 * sum(vals) / (3 * nvar) / 1e9, len(vals), nvar))
PY
