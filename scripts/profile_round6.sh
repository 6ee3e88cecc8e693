#!/bin/bash
# Round-6 evidence on the GPU box, ONE run from the final tree (every step bounded, stdin closed).  What it leaves in gpurun_out/r06/
# (scripts/collect_round6.sh copies the summaries into the tracked profiles/r06_*):
#   bench lines of all workloads (roofline, roofline_fit, roofline_rff, cpu_baseline, parity incl. parity.nontrivial), the default
#   model at scale (--ensemble 10), the per-rank share of an 8-GPU run, the N > 1 code paths on the one GPU, rocprofv3 kernel-trace
#   summaries, PMC passes (L2 -> fabric traffic of the sweep kernel at the N = 8192 and config-B geometries; MFMA busy of the
#   sweep kernel, rounds 5 and 6; MFMA busy of the factorisation), the factorisation series / soak, the sweep-schedule A/B with
#   the traffic of every schedule, profiles/r06_roofline.json.
# Usage (via gpurun): bash scripts/profile_round6.sh [quick]      then, in the build container: bash scripts/collect_round6.sh
set -u
export GPX_ROUND=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
QUICK=${1:-}
CPU=""; [ "$QUICK" = "quick" ] && CPU="--cpu-candidates 8192"
# FIRST, because the bench lines quote it (roofline.traffic is a profile-time constant):
# PMC: L2 -> fabric traffic of the sweep kernel (separate passes, --kernel-trace only): N = 8192 and config B's geometry
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --warm-steps 0 --plugin-steps 0 --no-refine --candidates 131072 < /dev/null > $O/pmc_$c.log 2>&1
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcb_$c -o p -- python $R/bench.py --workload b --steps 1 --warmup 0 --no-cpu-baseline --warm-steps 0 --plugin-steps 0 --no-refine --candidates 131072 < /dev/null > $O/pmcb_$c.log 2>&1
done
f1=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f1" ] && [ -n "$f2" ]; then python $R/scripts/pmc_traffic.py $f1 $f2 8192 65536 > $O/pmc_traffic.json; grep "k_sweep_trmm\|k_cross_gram" $f1 > $O/pmc_fetch_size.csv; grep "k_sweep_trmm\|k_cross_gram" $f2 > $O/pmc_write_size.csv; fi
f1=$(find $O/pmcb_FETCH_SIZE -name "*counter_collection.csv" | head -1); f2=$(find $O/pmcb_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f1" ] && [ -n "$f2" ]; then python $R/scripts/pmc_traffic.py $f1 $f2 2048 131072 > $O/pmc_traffic_b.json; fi
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmcb_FETCH_SIZE $O/pmcb_WRITE_SIZE
mkdir -p $R/profiles; for f in pmc_traffic.json pmc_traffic_b.json; do [ -s $O/$f ] && cp $O/$f $R/profiles/r06_$f; done      # roofline.traffic of the bench lines below reads these
cd $R
timeout 1200 python bench.py --steps 5 --warmup 1 $CPU < /dev/null > $O/bench_ns.json 2> $O/bench_ns.err
timeout 900 python bench.py --workload b --steps 5 --warmup 1 $CPU < /dev/null > $O/bench_b.json 2> $O/bench_b.err
timeout 900 python bench.py --workload c --steps 3 --warmup 1 $CPU < /dev/null > $O/bench_c.json 2> $O/bench_c.err
for wl in d e; do
  timeout 600 python bench.py --workload $wl --steps 3 --warmup 1 $CPU < /dev/null > $O/bench_$wl.json 2> $O/bench_$wl.err
done
timeout 1200 python bench.py --ensemble 10 --steps 2 --warmup 1 < /dev/null > $O/bench_ns_ens10.json 2> $O/bench_ns_ens10.err
# the per-rank share of an 8-GPU run on one GPU (2^17 candidates): what the replicated fit costs there
timeout 600 python bench.py --candidates 131072 --steps 5 --warmup 1 --no-cpu-baseline --plugin-steps 0 --no-refine --warm-steps 0 < /dev/null > $O/bench_ns_share8.json 2> $O/bench_ns_share8.err
mkdir -p $R/profiles; for wl in ns b c d e; do [ -s $O/bench_$wl.json ] && cp $O/bench_$wl.json $R/profiles/r06_bench_$wl.json; done      # (what the N > 1 lines' scaling_model reads)
# N > 1 code paths, dry runs on the one GPU (gloo transport, every rank on device 0; the RCCL launch must fail fast and loudly)
timeout 600 python bench.py --gpus 2 --backend gloo --share-device 0 --steps 3 --warmup 1 --no-refine --plugin-steps 0 --cpu-candidates 8192 < /dev/null > $O/bench_ns_2ranks_gloo_shared_gpu.json 2> $O/bench_ns_2ranks.err
timeout 600 python bench.py --gpus 2 --backend gloo --share-device 0 --workload d --steps 2 --warmup 1 --cpu-candidates 8192 < /dev/null > $O/bench_d_2ranks_gloo_shared_gpu.json 2> $O/bench_d_2ranks.err
( timeout 300 python bench.py --gpus 2 --steps 1 --no-cpu-baseline < /dev/null; echo "exit code: $?" ) > $O/bench_gpus2_rccl_on_one_gpu.log 2>&1
# kernel traces
TOPN=24 timeout 420 bash scripts/trace.sh ns --steps 2 --warmup 1 --warm-steps 2 --plugin-steps 0 < /dev/null > $O/trace_ns.txt 2>&1
TOPN=24 timeout 420 bash scripts/trace.sh b --workload b --steps 2 --warmup 1 --warm-steps 0 --plugin-steps 0 < /dev/null > $O/trace_b.txt 2>&1
TOPN=24 timeout 420 bash scripts/trace.sh d --workload d --steps 2 --warmup 1 < /dev/null > $O/trace_d.txt 2>&1
TOPN=24 timeout 420 bash scripts/trace.sh e --workload e --steps 2 --warmup 1 < /dev/null > $O/trace_e.txt 2>&1
# PMC: MFMA busy of the sweep kernel: round 5's schedule (tile_order 27) and round 6's (19)
cd /tmp && export TMPDIR=/tmp
mkdir -p $O/pmc_sq
for v in 27 19; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq/v$v -o p -- python $R/scripts/pmc_sweep.py $v < /dev/null > $O/pmc_sq/v$v.log 2>&1
  f=$(find $O/pmc_sq/v$v -name "*counter_collection.csv" | head -1)
  echo "## tile_order $v"; [ -n "$f" ] && python $R/scripts/pmc_parse.py $f | grep "k_sweep_trmm"; [ -n "$f" ] && grep "k_sweep_trmm" $f > $O/pmc_sq_v$v.csv
  rm -rf $O/pmc_sq/v$v
done > $O/pmc_sq_summary.txt 2>&1
# PMC: the task-graph factorisation at N = 16384 and 8192
for n in 16384 8192; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_chol_$n -o p -- python $R/scripts/tg/tg_sweep.py $n chol_tg=1 < /dev/null > $O/pmc_chol_$n.log 2>&1
  f=$(find $O/pmc_chol_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_parse.py $f | grep -i "chol_tg" > $O/pmc_chol_$n.txt
  rm -rf $O/pmc_chol_$n
done
cd $R
export GPX_DIAGNOSTICS=1            # the series below use the diagnostic knobs (traces, chunk lists)
# the factorisation by its own clock, sizes, the stream schedule beside it; factor + inverse
{
  for n in 2048 8192 16384; do timeout 200 python scripts/tg/tg_trace.py $n; echo; done
  for n in 256 512 1024 1536 2048 3001 4096 5000 8192 12288 14336 16384; do timeout 300 python scripts/tg/tg_sweep.py $n chol_tg=0 chol_tg=1; done
  for n in 4096 8192 12288 16384; do timeout 300 python scripts/ab/trtri_time.py $n trtri_ahead=0; timeout 300 python scripts/ab/trtri_time.py $n trtri_ahead=1; done
  echo; echo "# PMC (rocprofv3 --pmc, k_chol_tg launches of scripts/tg/tg_sweep.py):"
  for n in 16384 8192; do echo "## N = $n"; cat $O/pmc_chol_$n.txt 2>/dev/null; done
} > $O/chol_taskgraph.txt 2>&1
# the factorisation's soak: random sizes and many repetitions, every factor compared bit for bit
{ timeout 700 python scripts/tg/tg_fuzz_sizes.py 80 1; timeout 800 python scripts/tg/tg_soak.py 600; } > $O/chol_soak.txt 2>&1
# the sweep schedules side by side (stand-alone A/B binary: the library's kernels_sweep.hip compiled into it) and their L2 -> fabric traffic
( cd scripts/probe && ./sweep_ab.bin 8192 65536 5 19 27 15 7 31 23 11 19 27; ./sweep_ab.bin 2048 131072 5 19 27 15 7 ) > $O/sweep_schedules_ab.log 2>&1
bash scripts/probe/pmc_fetch.sh 8192 65536 19 18 17 16 19:4 19:16 27 15 7 > $O/sweep_schedules_traffic.txt 2>&1
timeout 300 python scripts/loglik_rate.py > $O/loglik_rate.txt 2>&1
unset GPX_DIAGNOSTICS
# the GPU test-suite of the same tree (the driver runs it again on its own box)
timeout 1500 python -m pytest tests -q -m gpu < /dev/null > $O/gpu_suite.log 2>&1
python scripts/r06_roofline.py $O > $O/roofline.json 2> $O/roofline.err
cut -c1-400 $O/bench_ns.json; echo; cat $O/pmc_traffic.json 2>/dev/null | head -30; tail -5 $O/bench_gpus2_rccl_on_one_gpu.log; head -40 $O/roofline.json
