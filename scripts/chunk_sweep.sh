#!/bin/bash
for c in 65536 131072 262144 524288 1048576; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --chunk $c 2>/dev/null > /tmp/b_$c.json
  python -c "
import json; d=json.load(open('/tmp/b_$c.json'))
print($c, round(d['ms_per_step'],1), round(d['roofline']['achieved'],2), {k:round(v,1) for k,v in d['stage_ms_per_step_rank0'].items() if k in ('cross_gram','sweep_trmm')})"
done
