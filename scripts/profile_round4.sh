#!/bin/bash
# Round-4 evidence on the GPU box, ONE run from the final tree (every step bounded, stdin closed):
#   bench lines for all workloads (roofline, roofline_fit, roofline_rff, cpu_baseline on 2^17 candidates in two runs, parity),
#   the N > 1 code paths on the one GPU (2 gloo ranks, the one-process sharded mode, and the loud refusal of a 2-rank RCCL
#   launch on one device), rocprofv3 kernel-trace summaries, PMC passes (HBM traffic of the sweep kernel; MFMA busy of the
#   task-graph factorisation), the factorisation's own critical-path stamps, and profiles/r04_roofline.json.
# Usage (via gpurun): bash scripts/profile_round4.sh [quick]
set -u
export GPX_ROUND=r04
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
QUICK=${1:-}
CPU=""; [ "$QUICK" = "quick" ] && CPU="--cpu-candidates 8192"
timeout 900 python bench.py --steps 5 --warmup 1 $CPU < /dev/null > $O/bench_ns.json 2> $O/bench_ns.err
timeout 900 python bench.py --workload b --steps 5 --warmup 1 $CPU < /dev/null > $O/bench_b.json 2> $O/bench_b.err
timeout 900 python bench.py --workload c --steps 3 --warmup 1 $CPU < /dev/null > $O/bench_c.json 2> $O/bench_c.err
for wl in d e; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 $CPU < /dev/null > $O/bench_$wl.json 2> $O/bench_$wl.err
done
# the 1-GPU lines feed the scaling_model of the N > 1 lines
mkdir -p $R/profiles; for wl in ns b c d e; do [ -s $O/bench_$wl.json ] && cp $O/bench_$wl.json $R/profiles/r04_bench_$wl.json; done
# N > 1 code paths, dry runs on the one GPU
timeout 600 python bench.py --gpus 2 --backend gloo --share-device 0 --steps 3 --warmup 1 --no-refine --plugin-steps 0 --cpu-candidates 8192 < /dev/null > $O/bench_ns_2ranks_gloo_shared_gpu.json 2> $O/bench_ns_2ranks.err
timeout 600 python bench.py --gpus 2 --backend gloo --share-device 0 --workload d --steps 2 --warmup 1 --cpu-candidates 8192 < /dev/null > $O/bench_d_2ranks_gloo_shared_gpu.json 2> $O/bench_d_2ranks.err
timeout 600 python bench.py --mode sharded --gpus 2 --share-device 0 --steps 3 --warmup 1 --cpu-candidates 8192 < /dev/null > $O/bench_ns_sharded_2handles_shared_gpu.json 2> $O/bench_ns_sharded.err
( timeout 300 python bench.py --gpus 2 --steps 1 --no-cpu-baseline < /dev/null; echo "exit code: $?" ) > $O/bench_gpus2_rccl_on_one_gpu.log 2>&1
# kernel traces
TOPN=24 timeout 420 bash scripts/trace.sh ns --steps 2 --warmup 1 --warm-steps 2 --plugin-steps 0 < /dev/null > $O/trace_ns.txt 2>&1
TOPN=24 timeout 420 bash scripts/trace.sh d --workload d --steps 2 --warmup 1 < /dev/null > $O/trace_d.txt 2>&1
TOPN=24 timeout 420 bash scripts/trace.sh e --workload e --steps 2 --warmup 1 < /dev/null > $O/trace_e.txt 2>&1
# PMC: HBM traffic of the sweep kernel (separate passes, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --warm-steps 0 --plugin-steps 0 --no-refine --candidates 131072 < /dev/null > $O/pmc_$c.log 2>&1
done
f1=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f1" ] && [ -n "$f2" ]; then python $R/scripts/pmc_traffic.py $f1 $f2 > $O/pmc_traffic.json; cp $f1 $O/pmc_fetch_size.csv; cp $f2 $O/pmc_write_size.csv; fi
# PMC: the task-graph factorisation at N = 16384 (throughput-bound) and 8192
for n in 16384 8192; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_chol_$n -o p -- python $R/scripts/tg/tg_sweep.py $n chol_tg=1 < /dev/null > $O/pmc_chol_$n.log 2>&1
  f=$(find $O/pmc_chol_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_parse.py $f | grep -i "chol_tg" > $O/pmc_chol_$n.txt
done
cd $R
# the factorisation by its own clock
{
  for n in 2048 4096 8192 16384; do timeout 200 python scripts/tg/tg_trace.py $n; echo; done
  timeout 300 python scripts/tg/tg_tasklog.py 8192; echo
  timeout 300 python scripts/tg/tg_tasklog.py 16384; echo
  for n in 2048 4096 8192 12288 16384; do timeout 300 python scripts/tg/tg_sweep.py $n chol_tg=0 chol_tg=1; done
  echo; echo "# PMC (rocprofv3 --pmc, k_chol_tg launches of scripts/tg/tg_sweep.py):"
  for n in 16384 8192; do echo "## N = $n"; cat $O/pmc_chol_$n.txt 2>/dev/null; done
} > $O/chol_taskgraph.txt 2>&1
# predict-with-gradients: wall time per call of the three forms, the kernels of the one-pass form, and where a warm plug-in
# iteration spends its time (alone; beside a second live handle with the runtime's default 4 hardware queues and with 8)
{
  for o in "grad_form=1 grad_kernel=0" "grad_form=1 grad_kernel=1" "grad_form=2" ""; do timeout 200 python scripts/grad_call_probe.py 8192 8 $o; done
  timeout 200 python scripts/grad_call_probe.py 2048 2
  bash scripts/grad_rb_sweep.sh 8192 8 "grad_form=0"
} > $O/grad_forms.txt 2>&1
{
  echo "# alone"; timeout 300 python scripts/plugin_phases.py ns 12
  echo; echo "# a second handle alive (as in bench.py's process), GPU_MAX_HW_QUEUES=4 (the runtime's default)"
  GPU_MAX_HW_QUEUES=4 PHASES_PRE=1 timeout 300 python scripts/plugin_phases.py ns 12 | head -8
  echo; echo "# a second handle alive, GPU_MAX_HW_QUEUES=8 (pybo_amd's default)"
  PHASES_PRE=1 timeout 300 python scripts/plugin_phases.py ns 12 | head -8
  echo; echo "# one-pass against two-pass gradients through the plug-in layer (GPX_OPTIONS), same box"
  bash scripts/plugin_ab.sh 12
} > $O/plugin_phases.txt 2>&1
python scripts/r04_roofline.py $O > $O/roofline.json 2> $O/roofline.err
cut -c1-400 $O/bench_ns.json; echo; cat $O/pmc_traffic.json 2>/dev/null | head -30; tail -5 $O/bench_gpus2_rccl_on_one_gpu.log; head -40 $O/roofline.json
