#!/bin/bash
# Round-2 evidence on the GPU box (every step bounded, stdin closed): bench lines for all workloads, rocprofv3
# kernel-trace summaries, PMC passes.  Usage (via gpurun): bash scripts/profile_round2.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 5 --warmup 1 < /dev/null > $O/bench_ns.json 2> $O/bench_ns.err
timeout 900 python bench.py --workload b --steps 5 --warmup 1 < /dev/null > $O/bench_b.json 2> $O/bench_b.err
for wl in c d e; do
  timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline < /dev/null > $O/bench_$wl.json 2> $O/bench_$wl.err
done
TOPN=14 timeout 420 bash scripts/trace.sh ns --steps 2 --warmup 1 --warm-steps 2 < /dev/null > $O/trace_ns.txt 2>&1
TOPN=14 timeout 420 bash scripts/trace.sh d --workload d --steps 2 --warmup 1 < /dev/null > $O/trace_d.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --warm-steps 0 --candidates 131072 < /dev/null > $O/pmc_$c.log 2>&1
done
f1=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f1" ] && [ -n "$f2" ]; then python $R/scripts/pmc_traffic.py $f1 $f2 > $O/pmc_traffic.json; fi
timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/scripts/pmc_sweep.py 23 < /dev/null > $O/pmc_sq.log 2>&1
fs=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
if [ -n "$fs" ]; then python $R/scripts/pmc_parse.py $fs | grep -i "sweep_trmm" > $O/pmc_sq_summary.txt; fi
cut -c1-400 $O/bench_ns.json; cat $O/pmc_traffic.json 2>/dev/null | head -30; cat $O/pmc_sq_summary.txt 2>/dev/null
