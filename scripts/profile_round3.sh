#!/bin/bash
# Round-3 evidence on the GPU box (every step bounded, stdin closed): bench lines for all workloads (each now carries
# parity + plugin_step), rocprofv3 kernel-trace summaries, the kernel trace of WARM plug-in iterations, the N > 1 launch
# path started by bench.py itself, PMC passes for the sweep kernel's HBM traffic.  Usage (via gpurun): bash scripts/profile_round3.sh
set -u
export GPX_ROUND=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 5 --warmup 1 < /dev/null > $O/bench_ns.json 2> $O/bench_ns.err
timeout 900 python bench.py --workload b --steps 5 --warmup 1 < /dev/null > $O/bench_b.json 2> $O/bench_b.err
timeout 600 python bench.py --workload c --steps 3 --warmup 1 < /dev/null > $O/bench_c.json 2> $O/bench_c.err
for wl in d e; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 < /dev/null > $O/bench_$wl.json 2> $O/bench_$wl.err
done
# the driver's command shape for N > 1, dry run on the one GPU (gloo, both ranks on device 0)
timeout 600 python bench.py --gpus 2 --backend gloo --share-device 0 --steps 3 --warmup 1 --no-refine --cpu-candidates 8192 < /dev/null > $O/bench_ns_gpus2_selflaunch.json 2> $O/bench_ns_gpus2_selflaunch.err
TOPN=16 timeout 420 bash scripts/trace.sh ns --steps 2 --warmup 1 --warm-steps 2 --plugin-steps 0 < /dev/null > $O/trace_ns.txt 2>&1
TOPN=16 timeout 420 bash scripts/trace.sh d --workload d --steps 2 --warmup 1 < /dev/null > $O/trace_d.txt 2>&1
# kernel trace of the plug-in loop: 1 cold + 3 warm iterations through policy / solver / recommender (no k_sweep_trmm after the first)
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --output-format csv -d $O/trace_plugin -o plugin -- python $R/scripts/plugin_iter.py --iters 3 < /dev/null > $O/trace_plugin.log 2>&1
cd $R
python scripts/plugin_trace_summary.py $O/trace_plugin/plugin_kernel_trace.csv > $O/plugin_warm_kernel_trace.txt 2>&1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --warm-steps 0 --plugin-steps 0 --no-refine --candidates 131072 < /dev/null > $O/pmc_$c.log 2>&1
done
f1=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f1" ] && [ -n "$f2" ]; then python $R/scripts/pmc_traffic.py $f1 $f2 > $O/pmc_traffic.json; fi
cd $R
cut -c1-300 $O/bench_ns.json; echo; cat $O/plugin_warm_kernel_trace.txt | head -40; cat $O/pmc_traffic.json 2>/dev/null | head -12
