#!/bin/bash
# plug-in warm iteration A/B on one box: GPX_OPTIONS sets the engine options of the handles the plug-in layer creates
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for o in "" "grad_form=1,grad_kernel=0" "" "grad_form=1,grad_kernel=0"; do
  GPX_OPTIONS="$o" timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --plugin-steps ${1:-12} --warm-steps 8 --no-refine < /dev/null > /tmp/ab.json 2>/tmp/ab.err
  python - "$o" <<'PY'
import json, sys
r = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
p = r['plugin_step']
print('GPX_OPTIONS=%-28r plugin warm %.2f ms (each %s)  beside a 40 ms objective %.2f ms   engine warm step %.2f ms' % (
    sys.argv[1], p['warm_ms'], ' '.join('%.1f' % v for v in p['warm_ms_each']), p['warm_ms_beside_a_40_ms_objective'], r['warm_step']['ms_per_step']))
PY
done
