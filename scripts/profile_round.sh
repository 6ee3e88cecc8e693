#!/bin/bash
# Collect the round's evidence on the GPU box: bench lines + rocprofv3 kernel-trace stats + PMC traffic.
# Usage (via gpurun): bash scripts/profile_round.sh r01
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 5 --warmup 1 > $OUT/bench_ns.json 2> $OUT/bench_ns.err
python bench.py --workload b --steps 5 --warmup 1 > $OUT/bench_b.json 2> $OUT/bench_b.err
python bench.py --workload c --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c.json 2> $OUT/bench_c.err
python bench.py --workload d --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_d.json 2> $OUT/bench_d.err
python bench.py --workload e --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_e.json 2> $OUT/bench_e.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ns -o ns -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace_ns.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d -o d -- python $R/bench.py --workload d --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace_d.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --candidates 131072 > $OUT/pmc_$c.log 2>&1
done
python $R/scripts/pmc_parse.py $OUT/pmc_FETCH_SIZE/p_counter_collection.csv > $OUT/pmc_summary.txt
python $R/scripts/pmc_parse.py $OUT/pmc_WRITE_SIZE/p_counter_collection.csv >> $OUT/pmc_summary.txt
cat $OUT/bench_ns.json; cat $OUT/pmc_summary.txt | grep -E "sweep|cross"
